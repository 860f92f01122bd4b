#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_phase
python tests/probes/r05_mpc_ls_hist.py 32 100 2>&1 | tee gpurun_out/r05_phase/mpc_ls_hist_c5.txt
