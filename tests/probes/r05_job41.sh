#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_lsm
for i in 1 2; do for v in lsc lsq; do
ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/$v.so python bench.py --config C5 --mpc-steps 200 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 loop $v:', '%.4g' % j['value'], '%.1f ms' % j['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r05_lsm/ab_lsc_na.txt
