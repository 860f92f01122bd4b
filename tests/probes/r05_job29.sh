#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_phase
ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/prof.so python tests/probes/phase_prof.py C5 64 4 2>&1 | tee gpurun_out/r05_phase/phase_c5_64_w4.txt
ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/prof.so python tests/probes/phase_prof.py C3 1024 2 2>&1 | tee gpurun_out/r05_phase/phase_c3_1024_w2.txt
