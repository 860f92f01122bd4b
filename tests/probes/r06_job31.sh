#!/bin/bash
# round 6, job 31: flat staging of the fused pass with eight elements of z and dz per lane and batch (variant sb8; default six)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job31; O=gpurun_out/r06_job31
bash tests/probes/ab.sh "--steps 20 --warmup 8" sb8 2>&1 | tee $O/ab_sb8_c2.txt
bash tests/probes/ab.sh "--config C4 --steps 10 --warmup 4" sb8 2>&1 | tee $O/ab_sb8_c4.txt
