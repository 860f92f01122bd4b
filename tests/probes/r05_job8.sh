#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_job8; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gputest.txt 2>&1; tail -8 $O/gputest.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-400 $O/bench_default.json
python bench.py --config C3 --no-cpu-baseline > $O/bench_c3.json 2>/dev/null; cut -c1-200 $O/bench_c3.json
python bench.py --config C5 --mpc-steps 200 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2>/dev/null; cut -c1-200 $O/bench_c5.json
python bench.py --config Q2 --no-cpu-baseline --no-pmc > $O/bench_q2.json 2>/dev/null; cut -c1-200 $O/bench_q2.json
python bench.py --config Q4 --no-cpu-baseline --no-pmc --steps 5 --warmup 2 > $O/bench_q4.json 2>/dev/null; cut -c1-200 $O/bench_q4.json
