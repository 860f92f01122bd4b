#!/bin/bash
# GPU box: the gate's flat pass unrolled by four (variant gateu4) against the shipped library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 300 python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/gateu4.so > $O/r04_gateu_bitwise.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 20 --warmup 3 --config C2" gateu4 > $O/r04_ab_gateu_c2.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 20 --warmup 3 --config C3" gateu4 > $O/r04_ab_gateu_c3.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 3 --warmup 1 --config C5 --mpc-steps 100" gateu4 > $O/r04_ab_gateu_c5.txt 2>&1
cat $O/r04_gateu_bitwise.txt $O/r04_ab_gateu_c2.txt $O/r04_ab_gateu_c3.txt $O/r04_ab_gateu_c5.txt
