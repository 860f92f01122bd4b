#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_ab
bash tests/probes/r05_ab.sh addr_c2 "--steps 20 --warmup 8" addr
bash tests/probes/ab.sh "--config C3 --steps 20 --warmup 5" addr 2>&1 | tee gpurun_out/r05_ab/ab_addr_c3.txt
bash tests/probes/ab.sh "--config C5 --mpc-steps 100 --steps 3 --warmup 1" addr 2>&1 | tee gpurun_out/r05_ab/ab_addr_c5.txt
