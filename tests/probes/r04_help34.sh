#!/bin/bash
# GPU box: team of two, helper wavefront levels 2 / 3 / 4 (default): bitwise check against level 2, same-box A/B on C3 and the C3-shaped MPC loop
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 300 python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/help2.so > $O/r04_help34_bitwise.txt 2>&1
timeout 300 python tests/probes/bitwise_ab.py algames.jl_amd/lib/variants/help3.so algames.jl_amd/lib/variants/help2.so >> $O/r04_help34_bitwise.txt 2>&1
timeout 600 bash tests/probes/ab.sh "--steps 20 --warmup 3 --config C3" help3 help2 > $O/r04_ab_help34_c3.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_full_batch.py -x -q -k "team or c3" 2>&1 | tail -3 > $O/r04_help34_tests.txt
cat $O/r04_help34_bitwise.txt $O/r04_ab_help34_c3.txt $O/r04_help34_tests.txt
