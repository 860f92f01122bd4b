#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_ab
python tests/probes/r05_diff.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/pw.so 2>&1 | tee gpurun_out/r05_ab/diff_pw.txt
bash tests/probes/ab.sh "--config C5 --mpc-steps 100 --steps 3 --warmup 1" pw 2>&1 | tee gpurun_out/r05_ab/ab_pw_c5.txt
bash tests/probes/ab.sh "--config C2 --steps 20 --warmup 5 --games-per-gpu 512" pw 2>&1 | tee gpurun_out/r05_ab/ab_pw_c2s.txt
ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/pw.so timeout 900 python -m pytest tests/test_gpu_full_batch.py tests/test_gpu_refinement.py -m gpu -q -x 2>&1 | tail -4
