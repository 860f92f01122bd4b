#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_phase
for g in 256 4096; do
ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/prof.so python tests/probes/phase_prof.py C2 $g 2>&1 | tee gpurun_out/r05_phase/phase_c2_$g.txt
done
