#!/bin/bash
# round 6, job 44: opt_x trips of the lane-role dealing two behind one guard (variant pair; C2 unit): bit-identity, same-box A/B
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job44; O=gpurun_out/r06_job44
python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/pair.so 2>&1 | grep C2 | tee $O/bitwise_pair.txt
bash tests/probes/ab.sh "--steps 20 --warmup 8" pair 2>&1 | tee $O/ab_pair_c2.txt
bash tests/probes/ab.sh "--config C4 --steps 10 --warmup 4" pair 2>&1 | tee $O/ab_pair_c4.txt
