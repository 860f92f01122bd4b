#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_final2
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 | tee gpurun_out/r05_final2/gputest.txt
