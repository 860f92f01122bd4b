#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_lsm
for m in 0 1; do
ALGAMES_LS_MULTI=$m ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/prof.so python tests/probes/r05_mpc_prof.py 64 200 4 2>&1 | tee gpurun_out/r05_lsm/mpc_prof_c5_lsmulti$m.txt
done
timeout 900 python -m pytest tests/test_gpu_line_search_batch.py -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/r05_lsm/test.txt
for i in 1 2 3; do for m in 0 1; do
ALGAMES_LS_MULTI=$m python bench.py --config C5 --mpc-steps 200 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 loop, 64 seeds x 200 steps, ALGAMES_LS_MULTI=$m:', '%.4g' % j['value'], 'game-iterations/s', '%.1f ms' % j['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r05_lsm/ab_lsmulti_c5loop.txt
python tests/probes/hetero.py 2>&1 | head -24 | tee gpurun_out/r05_lsm/hetero.txt
