#!/bin/bash
# GPU box: team of four, y_i / g_c and the A' table on wavefronts 2 / 3 (default) against -DALG_TEAM_YSPLIT=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 300 python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/noysplit.so > $O/r04_ysplit_bitwise.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 3 --warmup 1 --config C5 --mpc-steps 100" noysplit > $O/r04_ab_ysplit_c5.txt 2>&1
timeout 300 bash tests/probes/ab.sh "--steps 20 --warmup 3 --config C2 --games-per-gpu 512" noysplit > $O/r04_ab_ysplit_c2s.txt 2>&1
timeout 300 bash tests/probes/ab.sh "--steps 20 --warmup 3 --config C5" noysplit > $O/r04_ab_ysplit_c5solve.txt 2>&1
cat $O/r04_ysplit_bitwise.txt $O/r04_ab_ysplit_c5.txt $O/r04_ab_ysplit_c2s.txt $O/r04_ab_ysplit_c5solve.txt
