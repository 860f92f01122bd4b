#!/bin/bash
# GPU box: 128-register kernels, coefficient groups of twelve (variant) against four (default); phase-cycle profiles of the current tree
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 600 bash tests/probes/ab.sh "--steps 20 --warmup 3 --config C2" rdgw4_12 > $O/r04_ab_rdgw4_c2.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 10 --warmup 3 --config C4" rdgw4_12 > $O/r04_ab_rdgw4_c4.txt 2>&1
bash tests/probes/phase_prof.sh run C3 1024 2 > $O/r04_phase2_c3_team2.txt 2>&1
bash tests/probes/phase_prof.sh run C5 64 4 > $O/r04_phase2_c5_team4.txt 2>&1
bash tests/probes/phase_prof.sh run C2 4096 1 > $O/r04_phase2_c2.txt 2>&1
cat $O/r04_ab_rdgw4_c2.txt $O/r04_ab_rdgw4_c4.txt; tail -13 $O/r04_phase2_c3_team2.txt; tail -13 $O/r04_phase2_c5_team4.txt; tail -13 $O/r04_phase2_c2.txt
