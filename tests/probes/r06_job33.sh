#!/bin/bash
# round 6, job 33: seed 101156 of the long base family (4-player unicycle, N = 14: 1 of 2000 outside the rule in job 32) on the shipped library and on the
# flat dealing of HEAD~ (variant norowidx: no carried indices, no lane roles); phase cycles of C2 on the profile build
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_job33; mkdir -p $O
python tests/probes/r06_seed_compare.py 101156 2>&1 | tee $O/seed_101156_shipped.txt
ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/norowidx.so python tests/probes/r06_seed_compare.py 101156 2>&1 | tee $O/seed_101156_norowidx.txt
ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/prof.so python tests/probes/phase_prof.py C2 4096 > $O/phase_cycles_c2_4096.txt 2>&1; cat $O/phase_cycles_c2_4096.txt
