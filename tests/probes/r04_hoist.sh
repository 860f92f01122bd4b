#!/bin/bash
# GPU box: LDS reads behind the FMA chains requested in front of them (default) against the tree before (variant head.so)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
V=${1:-head}
timeout 300 python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/$V.so > $O/r04_hoist_bitwise.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 20 --warmup 3 --config C3" $V > $O/r04_ab_hoist_c3.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 3 --warmup 1 --config C5 --mpc-steps 100" $V > $O/r04_ab_hoist_c5.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 20 --warmup 3 --config C2" $V > $O/r04_ab_hoist_c2.txt 2>&1
cat $O/r04_hoist_bitwise.txt $O/r04_ab_hoist_c3.txt $O/r04_ab_hoist_c5.txt $O/r04_ab_hoist_c2.txt
