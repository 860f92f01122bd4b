#!/bin/bash
# round 6, job 36: the outliers of the 4x long fuzz run as a committed test (oracle-sensitivity rule); all outliers of the extended family (the long script printed eight of ten);
# record pass staged in one batch (variant rec12) against two batches of six
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_job36; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -s -k "long_run_outliers" 2>&1 | grep -v "^arbiter consulted" | tail -40 | cut -c1-400 | tee $O/outliers_test.txt
bash tests/probes/ab.sh "--steps 20 --warmup 8" rec12 2>&1 | tee $O/ab_rec12_c2.txt
timeout 600 python tests/probes/fuzz_long_r6.py 1600 extended 2>&1 | cut -c1-120 | tee $O/fuzz_extended_1600_all.txt
