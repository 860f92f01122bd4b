#!/bin/bash
# round 6, job 18: receding-horizon loop kernels with laundered argument loads / solve set-up (SGPR spills of the C5 loop kernel 25 -> 13): GPU suite,
# the four BASELINE shapes' rates, resources of the shipped library
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job18; O=gpurun_out/r06_job18
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/gputest.txt
for spec in "C2 20 8" "C3 20 8" "C4 10 4"; do set -- $spec
  python bench.py --config $1 --steps $2 --warmup $3 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', '%.4g game-iterations/s' % d['value'], '%.3f ms' % d['ms_per_step'])" | tee -a $O/rates.txt
done
for i in 1 2; do python bench.py --config C5 --mpc-steps 200 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5 loop 64 x 200', '%.4g game-iterations/s' % d['value'], '%.3f ms' % d['ms_per_step'])" | tee -a $O/rates.txt; done
python bench.py --config C5 --mpc-steps 50 --games-per-gpu 4096 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5 loop 4096 x 50 (one wavefront per game)', '%.4g' % d['value'])" | tee -a $O/rates.txt
python -c "
import sys; sys.path.insert(0,'.')
import algames_jl_amd._resources as r
res=r.kernel_resources('algames.jl_amd/lib/libalgames_hip.so')
for k in sorted(res):
    if 'k_newton_solve' in k or 'k_mpc_loop' in k or 'k_newton_resume' in k: print(k, res[k])
" > $O/kernel_resources.txt
