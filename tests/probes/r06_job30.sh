#!/bin/bash
# round 6, job 30: lane roles in the staging of the fused pass too (ALG_R6_LANEROLE 2, four blocks per batch) against rows only (variant lr1)
# and five blocks per batch (variant lr2sb5): bit-identity (the staging moves no value), same-box A/B on C2 / C4
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job30; O=gpurun_out/r06_job30
python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/lr1.so 2>&1 | tail -8 | tee $O/bitwise_lr2_vs_lr1.txt
bash tests/probes/ab.sh "--steps 20 --warmup 8" lr1 lr2sb5 2>&1 | tee $O/ab_lr2_c2.txt
bash tests/probes/ab.sh "--config C4 --steps 10 --warmup 4" lr1 2>&1 | tee $O/ab_lr2_c4.txt
