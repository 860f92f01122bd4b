#!/bin/bash
# whole GPU suite on the split-recursion build, A/B against the round-4 form, phase cycles
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_job3; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputest.txt 2>&1; tail -15 $O/gputest.txt
bash tests/probes/ab.sh "--steps 20 --warmup 5" nosplit > $O/ab_c2.txt 2>&1; cat $O/ab_c2.txt
for g in 256 4096; do ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/prof.so python tests/probes/phase_prof.py C2 $g 1 > $O/phase_c2_$g.txt 2>&1; cat $O/phase_c2_$g.txt; done
