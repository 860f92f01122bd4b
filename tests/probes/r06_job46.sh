#!/bin/bash
# round 6, job 46: sweep prefetch depth of the 256-register kernels 2 (default) against 4 (variant sdw4): bit-identity, C3 / C5 loop / large-batch unicycle shapes
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job46; O=gpurun_out/r06_job46
python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/sdw4.so 2>&1 | tail -6 | tee $O/bitwise_sdw4.txt
bash tests/probes/ab.sh "--config C3 --steps 20 --warmup 8" sdw4 2>&1 | tee $O/ab_sdw4_c3.txt
bash tests/probes/ab.sh "--config C5 --mpc-steps 200 --steps 2 --warmup 1" sdw4 2>&1 | tee $O/ab_sdw4_c5loop.txt
bash tests/probes/ab.sh "--config C5 --games-per-gpu 4096 --steps 10 --warmup 4" sdw4 2>&1 | tee $O/ab_sdw4_c5_4096.txt
bash tests/probes/ab.sh "--config C3 --games-per-gpu 4096 --steps 10 --warmup 4" sdw4 2>&1 | tee $O/ab_sdw4_c3_4096.txt
