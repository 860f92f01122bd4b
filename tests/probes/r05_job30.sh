#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_phase
ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/prof.so python tests/probes/r05_mpc_prof.py 64 200 4 2>&1 | tee gpurun_out/r05_phase/mpc_prof_c5.txt
