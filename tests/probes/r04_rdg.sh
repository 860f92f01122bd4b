#!/bin/bash
# GPU box: row-broadcast FMA chains with their coefficients requested in groups (default: 256-register kernels; rdgw4_4: also the
# 128-register kernels, groups of four) against fetch-where-used (rdg0)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 300 python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/rdg0.so > $O/r04_rdg_bitwise.txt 2>&1
timeout 300 python tests/probes/bitwise_ab.py algames.jl_amd/lib/variants/rdgw4_4.so algames.jl_amd/lib/variants/rdg0.so >> $O/r04_rdg_bitwise.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 20 --warmup 3 --config C3" rdg0 > $O/r04_ab_rdg_c3.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 3 --warmup 1 --config C5 --mpc-steps 100" rdg0 > $O/r04_ab_rdg_c5.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 20 --warmup 3 --config C2" rdgw4_4 > $O/r04_ab_rdg_c2.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 20 --warmup 3 --config C2 --games-per-gpu 512" rdg0 > $O/r04_ab_rdg_c2s.txt 2>&1
cat $O/r04_rdg_bitwise.txt $O/r04_ab_rdg_c3.txt $O/r04_ab_rdg_c5.txt $O/r04_ab_rdg_c2.txt $O/r04_ab_rdg_c2s.txt
