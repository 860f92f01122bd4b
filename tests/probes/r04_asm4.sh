#!/bin/bash
# GPU box: four terms of a row-broadcast chain per asm statement (default) against one (noasm4)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 300 python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/noasm4.so > $O/r04_asm4_bitwise.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 20 --warmup 3 --config C3" noasm4 > $O/r04_ab_asm4_c3.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 3 --warmup 1 --config C5 --mpc-steps 100" noasm4 > $O/r04_ab_asm4_c5.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 20 --warmup 3 --config C2" noasm4 > $O/r04_ab_asm4_c2.txt 2>&1
cat $O/r04_asm4_bitwise.txt $O/r04_ab_asm4_c3.txt $O/r04_ab_asm4_c5.txt $O/r04_ab_asm4_c2.txt
