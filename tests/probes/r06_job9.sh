#!/bin/bash
# round 6, job 9: seven to nine players + d = 1 (parity, fuzz, guard zones), dense gate tolerance (tests + Q2 / Q4 A/B), then the suite
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_job9; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_ext.py "tests/test_gpu_fuzz.py::test_fuzz_seven_to_nine_players" "tests/test_gpu_fuzz.py::test_fuzz_long_run_seeds_of_round_5" "tests/test_gpu_fuzz.py::test_direction_backward_error_against_the_arbiter" -q 2>&1 | grep "Error\|^E  \|passed\|failed\|^FAILED" | cut -c1-500 > $O/tests_breadth.txt; tail -15 $O/tests_breadth.txt
for cfgspec in "Q2 4096" "Q4 1024"; do set -- $cfgspec
  bash tests/probes/ab.sh "--config $1 --games-per-gpu $2 --steps 8 --warmup 3" dense_tol8 dense_tol6 > $O/ab_dense_tol_$1.txt 2>&1; cat $O/ab_dense_tol_$1.txt
done
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/gputest.txt; tail -6 $O/gputest.txt | cut -c1-300
