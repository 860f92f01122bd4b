#!/bin/bash
# round 6, job 21: row-broadcast half-chains as the default: GPU suite, same-box A/B against the single chains (variant rdone: C2 / C3 / team kernels),
# default and driver-style bench lines
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job21; O=gpurun_out/r06_job21
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^arbiter consulted\|^forward error\|^decision\|^status differs\|^device omega" | tail -12 > $O/gputest.txt; grep -E "forced corrections|passed|failed|FAILED" $O/gputest.txt | cut -c1-300
bash tests/probes/ab.sh "--steps 20 --warmup 8" rdone 2>&1 | tee $O/ab_rdone_c2.txt
bash tests/probes/ab.sh "--config C3 --steps 20 --warmup 8" rdone 2>&1 | tee $O/ab_rdone_c3.txt
bash tests/probes/ab.sh "--config C5 --mpc-steps 200 --steps 2 --warmup 1" rdone 2>&1 | tee $O/ab_rdone_c5loop.txt
unset ALGAMES_HIP_LIB
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']; print('C2 default', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], 'frac %.4f' % r['frac'], 'traffic %.4g' % (r.get('traffic') or 0), 'over model', r.get('traffic_over_model'))"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $O/bench_steps20_warmup5.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_steps20_warmup5.json').read().strip().splitlines()[-1]); print('C2 --steps 20 --warmup 5', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'])"
