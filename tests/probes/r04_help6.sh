#!/bin/bash
# GPU box: team of two, t_i / s_i in registers (default, ALG_HELP2 = 6) against level 4
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 300 python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/help4.so > $O/r04_help6_bitwise.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 20 --warmup 3 --config C3" help4 > $O/r04_ab_help6_c3.txt 2>&1
cat $O/r04_help6_bitwise.txt $O/r04_ab_help6_c3.txt
