#!/bin/bash
# round 6, job 29: fused pass of the double integrator with lane roles (ALG_R6_LANEROLE: a lane keeps its row of the step, trips unrolled over the
# steps) against the flat dealing with carried indices (variant nolr = LANEROLE 0, ROWIDX 1) and the flat dealing of HEAD (variant norowidx).
# Parity tests of the base configurations, same-box A/B on C2 / C4 / C5 loop (the unicycle kernels differ between default and norowidx only by ROWIDX)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job29; O=gpurun_out/r06_job29
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_batch.py tests/test_gpu_boundary.py -m gpu -q -x 2>&1 | tail -4 | tee $O/gputest_subset.txt
bash tests/probes/ab.sh "--steps 20 --warmup 8" nolr norowidx 2>&1 | tee $O/ab_lanerole_c2.txt
bash tests/probes/ab.sh "--config C4 --steps 10 --warmup 4" nolr 2>&1 | tee $O/ab_lanerole_c4.txt
bash tests/probes/ab.sh "--config C5 --mpc-steps 200 --steps 2 --warmup 1" norowidx 2>&1 | tee $O/ab_rowidx_c5loop.txt
bash tests/probes/ab.sh "--config C3 --games-per-gpu 4096 --steps 10 --warmup 4" norowidx 2>&1 | tee $O/ab_rowidx_c3_4096.txt
