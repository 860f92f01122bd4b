#!/bin/bash
# round 6, job 19: settle_traj as one copy instead of an exchange -- GPU suite, default bench line (traffic from the in-run PMC passes), other shapes
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job19; O=gpurun_out/r06_job19
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/gputest.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']; print('C2 default', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'], 'frac %.4f' % r['frac'], 'traffic %.4g' % (r.get('traffic') or 0), 'over model', r.get('traffic_over_model'))"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $O/bench_steps20_warmup5.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_steps20_warmup5.json').read().strip().splitlines()[-1]); print('C2 --steps 20 --warmup 5', '%.4g' % d['value'], '%.3f ms' % d['ms_per_step'])"
for spec in "C3 20 8" "C4 10 4"; do set -- $spec
  python bench.py --config $1 --steps $2 --warmup $3 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', '%.4g game-iterations/s' % d['value'], '%.3f ms' % d['ms_per_step'])" | tee -a $O/rates.txt
done
python bench.py --config C5 --mpc-steps 200 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5 loop 64 x 200', '%.4g game-iterations/s' % d['value'], '%.3f ms' % d['ms_per_step'])" | tee -a $O/rates.txt
