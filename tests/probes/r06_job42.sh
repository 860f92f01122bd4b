#!/bin/bash
# round 6, job 42: lane roles in the fused pass of the one-wavefront unicycle kernels, the line search's group pass dealing its rows the same way
# (ALG_R6_LANEROLE_UNI; variant nolru = off).  Bitwise tests of the group pass, unicycle parity, same-box A/B on the large-batch unicycle shapes
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job42; O=gpurun_out/r06_job42
timeout 900 python -m pytest tests/test_gpu_line_search_batch.py -m gpu -q -x 2>&1 | tail -3 | tee $O/gputest_ls_batch.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_batch.py -m gpu -q -x -k "unicycle or c5 or c3 or C5 or C3 or mpc" 2>&1 | tail -3 | tee $O/gputest_unicycle.txt
bash tests/probes/ab.sh "--config C5 --games-per-gpu 4096 --steps 10 --warmup 4" nolru 2>&1 | tee $O/ab_lru_c5_4096.txt
bash tests/probes/ab.sh "--config C3 --games-per-gpu 4096 --steps 10 --warmup 4" nolru 2>&1 | tee $O/ab_lru_c3_4096.txt
bash tests/probes/ab.sh "--config C5 --games-per-gpu 1024 --steps 10 --warmup 4" nolru 2>&1 | tee $O/ab_lru_c5_1024.txt
bash tests/probes/ab.sh "--config C5 --mpc-steps 50 --games-per-gpu 4096 --steps 2 --warmup 1" nolru 2>&1 | tee $O/ab_lru_c5loop_4096.txt
