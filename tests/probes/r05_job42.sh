#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_wp
V=${1:-wp}
python tests/probes/bitwise_ab.py algames.jl_amd/lib/variants/$V.so 2>&1 | tail -8 | tee gpurun_out/r05_wp/bitwise_$V.txt
python tests/probes/r05_diff.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/$V.so 2>&1 | tail -8 | tee gpurun_out/r05_wp/diff_$V.txt
for i in 1 2 3; do for v in default $V; do
L=$R/algames.jl_amd/lib/variants/$v.so; [ $v = default ] && L=$R/algames.jl_amd/lib/libalgames_hip.so
ALGAMES_HIP_LIB=$L python bench.py --config C5 --mpc-steps 200 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 loop $v:', '%.4g' % j['value'], '%.1f ms' % j['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r05_wp/ab_${V}_c5loop.txt
for c in "C5 --games-per-gpu 256" "C3 --games-per-gpu 128"; do for v in default $V; do
L=$R/algames.jl_amd/lib/variants/$v.so; [ $v = default ] && L=$R/algames.jl_amd/lib/libalgames_hip.so
ALGAMES_HIP_LIB=$L python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $v:', '%.4g' % j['value'], j['ms_per_step'], j['config']['wavefronts_per_game'])"
done; done 2>&1 | tee gpurun_out/r05_wp/ab_${V}_other.txt
