#!/bin/bash
# final evidence of round 5 (second half: grouped line search, chunk size 10) on one box: the whole GPU suite, the evidence run (r05_measure.sh),
# the receding-horizon loop's pass accounts, the long fuzz run
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v "^arbiter consulted\|^forward error\|^decision" | tail -12 > gpurun_out/r05_gputest_final.txt; cat gpurun_out/r05_gputest_final.txt
bash tests/probes/r05_measure.sh > gpurun_out/r05_measure.log 2>&1; tail -3 gpurun_out/r05_measure.log | cut -c1-300
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > gpurun_out/bench_r05_steps20_warmup5.json 2>/dev/null
python tests/probes/hetero.py > gpurun_out/r05_hetero.txt 2>&1
python tests/probes/fuzz_long_r4.py 400 > gpurun_out/r05_fuzz_long_final.txt 2>&1; cut -c1-200 gpurun_out/r05_fuzz_long_final.txt
