#!/bin/bash
# round 6 evidence job on the final binary: GPU suite, the measurements of r06_measure.sh, the long fuzz run
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^arbiter consulted\|^forward error\|^decision\|^status differs\|^device omega" | tail -12 > gpurun_out/r06_gputest_final.txt; tail -3 gpurun_out/r06_gputest_final.txt
bash tests/probes/r06_measure.sh > gpurun_out/r06_measure.log 2>&1; tail -25 gpurun_out/r06_measure.log | cut -c1-250
timeout 1200 python tests/probes/fuzz_long_r6.py 400 > gpurun_out/r06_fuzz_long_final.txt 2>&1; cut -c1-330 gpurun_out/r06_fuzz_long_final.txt
