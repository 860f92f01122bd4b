#!/bin/bash
# round 6, job 25: per-step base addresses of the sweeps' global accesses in scalar registers (ALG_R6_SADDR; variant nosaddr = without): GPU suite,
# same-box A/B on C2 / C3 / C5 loop
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job25; O=gpurun_out/r06_job25
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/gputest.txt
bash tests/probes/ab.sh "--steps 20 --warmup 8" nosaddr 2>&1 | tee $O/ab_saddr_c2.txt
bash tests/probes/ab.sh "--config C3 --steps 20 --warmup 8" nosaddr 2>&1 | tee $O/ab_saddr_c3.txt
bash tests/probes/ab.sh "--config C5 --mpc-steps 200 --steps 2 --warmup 1" nosaddr 2>&1 | tee $O/ab_saddr_c5loop.txt
