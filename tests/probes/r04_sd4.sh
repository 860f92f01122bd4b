#!/bin/bash
# GPU box: prefetch ring of the forward / costate sweeps four steps deep in the 256-register kernels (variant sd4) against two (shipped)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 300 python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/sd4.so > $O/r04_sd4_bitwise.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 20 --warmup 3 --config C3" sd4 > $O/r04_ab_sd4_c3.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 3 --warmup 1 --config C5 --mpc-steps 100" sd4 > $O/r04_ab_sd4_c5.txt 2>&1
cat $O/r04_sd4_bitwise.txt $O/r04_ab_sd4_c3.txt $O/r04_ab_sd4_c5.txt
