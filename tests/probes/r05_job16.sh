#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_ab
export ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/splitu.so
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_batch.py tests/test_gpu_refinement.py tests/test_gpu_parity_ext.py tests/test_gpu_boundary.py -m gpu -x -q 2>&1 | tail -12
unset ALGAMES_HIP_LIB
bash tests/probes/ab.sh "--config C5 --mpc-steps 100 --steps 3 --warmup 1" splitu 2>&1 | tee gpurun_out/r05_ab/ab_splitu_c5.txt
bash tests/probes/ab.sh "--config C5 --steps 20 --warmup 5 --games-per-gpu 4096" splitu 2>&1 | tee gpurun_out/r05_ab/ab_splitu_c5big.txt
