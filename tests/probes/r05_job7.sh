#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
bash tests/probes/r05_ab.sh nolso_c3 "--config C3 --steps 20 --warmup 5" nolso
bash tests/probes/ab.sh "--config C5 --mpc-steps 100 --steps 3 --warmup 1" nolso 2>&1 | tee gpurun_out/r05_ab/ab_nolso_c5.txt
bash tests/probes/ab.sh "--config C2 --steps 20 --warmup 5 --games-per-gpu 512" nolso 2>&1 | tee gpurun_out/r05_ab/ab_nolso_c2s.txt
bash tests/probes/ab.sh "--config C2 --steps 20 --warmup 5" nolso 2>&1 | tee gpurun_out/r05_ab/ab_nolso_c2.txt
