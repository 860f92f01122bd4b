#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_ab
bash tests/probes/ab.sh "--config C3 --steps 20 --warmup 5" split4 2>&1 | tee gpurun_out/r05_ab/ab_split4_c3.txt
python tests/probes/r05_diff.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/split4.so 2>&1 | tee gpurun_out/r05_ab/diff_split4.txt
ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/split4.so timeout 1500 python -m pytest tests/test_gpu_full_batch.py tests/test_gpu_refinement.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -6
