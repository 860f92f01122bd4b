#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_ow
V=${1:-ow}
ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/$V.so timeout 900 python -m pytest tests/test_gpu_line_search_batch.py -m gpu -q 2>&1 | grep "FAILED\|passed\|failed\|assert" | head -12
for i in 1 2; do for m in 0 1; do
ALGAMES_LS_MULTI=$m ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/$V.so python bench.py --config C5 --mpc-steps 50 --games-per-gpu 4096 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 loop 4096 seeds x 50 steps, one wavefront per game, ALGAMES_LS_MULTI=$m:', '%.4g' % j['value'], '%.1f ms' % j['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r05_ow/ab_c5loop_4096.txt
for c in "C5 --games-per-gpu 1024" "C5 --games-per-gpu 4096" "C3 --games-per-gpu 4096"; do for v in default $V; do
L=$R/algames.jl_amd/lib/variants/$v.so; [ $v = default ] && L=$R/algames.jl_amd/lib/libalgames_hip.so
ALGAMES_HIP_LIB=$L python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $v:', '%.4g' % j['value'], j['ms_per_step'], j['config']['wavefronts_per_game'])"
done; done 2>&1 | tee gpurun_out/r05_ow/ab_other.txt
