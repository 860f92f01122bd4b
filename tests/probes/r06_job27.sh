#!/bin/bash
# round 6, job 27: forward / costate sweeps -- |du| + |dx| summed and checked without per-step lane masks (ALG_R6_PL1), LDS double-buffer index as a
# compile-time constant of the unrolled step (ALG_R6_CURC); variant nopc = both off.  GPU suite, same-box A/B on C2 / C3 / C5 loop
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job27; O=gpurun_out/r06_job27
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/gputest.txt
bash tests/probes/ab.sh "--steps 20 --warmup 8" nopc 2>&1 | tee $O/ab_pl1_curc_c2.txt
bash tests/probes/ab.sh "--config C3 --steps 20 --warmup 8" nopc 2>&1 | tee $O/ab_pl1_curc_c3.txt
bash tests/probes/ab.sh "--config C5 --mpc-steps 200 --steps 2 --warmup 1" nopc 2>&1 | tee $O/ab_pl1_curc_c5loop.txt
