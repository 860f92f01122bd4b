#!/bin/bash
# round 6, job 22: row-broadcast half-chains on the double-integrator kernels only (the default): GPU suite, same-box A/B against the single chains
# (variant rdone: C2 kernels and the team kernels), C3 / C5 loop rates (unicycle kernels: unchanged arithmetic), default and driver-style bench lines
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job22; O=gpurun_out/r06_job22
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/gputest.txt
bash tests/probes/ab.sh "--steps 20 --warmup 8" rdone 2>&1 | tee $O/ab_rdone_c2.txt
bash tests/probes/ab.sh "--config C4 --steps 10 --warmup 4" rdone 2>&1 | tee $O/ab_rdone_c4.txt
bash tests/probes/ab.sh "--games-per-gpu 512 --steps 20 --warmup 8" rdone 2>&1 | tee $O/ab_rdone_c2_512.txt
unset ALGAMES_HIP_LIB
for i in 1 2; do python bench.py --config C5 --mpc-steps 200 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5 loop 64 x 200', '%.4g game-iterations/s' % d['value'], '%.3f ms' % d['ms_per_step'])" | tee -a $O/rates.txt; done
python bench.py --config C3 --steps 20 --warmup 8 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3', '%.4g game-iterations/s' % d['value'], '%.3f ms' % d['ms_per_step'])" | tee -a $O/rates.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']; print('C2 default', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], 'frac %.4f' % r['frac'], 'traffic %.4g' % (r.get('traffic') or 0), 'over model', r.get('traffic_over_model'))" | tee -a $O/rates.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $O/bench_steps20_warmup5.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_steps20_warmup5.json').read().strip().splitlines()[-1]); print('C2 --steps 20 --warmup 5', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'])" | tee -a $O/rates.txt
