#!/bin/bash
# round 6, job 40: rows per lane and pass of the assemble row loops in the team kernels (256 registers): 2 (default) against 3 / 4 (variants ur3 / ur4, unit mw)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job40; O=gpurun_out/r06_job40
for v in ur3 ur4; do echo $v; python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/$v.so 2>&1 | tail -6; done | tee $O/bitwise_ur.txt
bash tests/probes/ab.sh "--config C3 --steps 20 --warmup 8" ur3 ur4 2>&1 | tee $O/ab_ur_c3.txt
bash tests/probes/ab.sh "--config C5 --mpc-steps 200 --steps 2 --warmup 1" ur3 ur4 2>&1 | tee $O/ab_ur_c5loop.txt
bash tests/probes/ab.sh "--games-per-gpu 512 --steps 20 --warmup 8" ur4 2>&1 | tee $O/ab_ur_c2_512.txt
