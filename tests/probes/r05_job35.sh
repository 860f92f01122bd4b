#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_lsm
for i in 1 2; do for v in default lsmb6 lsmb8; do
L=$R/algames.jl_amd/lib/variants/$v.so; [ $v = default ] && L=$R/algames.jl_amd/lib/libalgames_hip.so
ALGAMES_HIP_LIB=$L python bench.py --config C5 --mpc-steps 200 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 loop $v:', '%.4g' % j['value'], j['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r05_lsm/ab_c5loop_na_hoisted.txt
for c in "C3" "C5 --games-per-gpu 1024" "C2 --games-per-gpu 512"; do for m in 0 1; do
ALGAMES_LS_MULTI=$m python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c ls_multi=$m:', '%.4g' % j['value'], j['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r05_lsm/ab_other_shapes.txt
