#!/bin/bash
# round 6, job 15: ten players (TIGHT LDS layout of the dense direction) -- GPU suite on the rebuilt binary, solve rates of five to ten players,
# the default bench line (the tile-path kernels must be untouched), the ten-player family of the long fuzz generator
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job15
timeout 1500 python -m pytest tests -m gpu -q -s -x 2>&1 | grep -v "^arbiter consulted\|^forward error\|^decision\|^status differs\|^device omega" | tail -15 > gpurun_out/r06_job15/gputest.txt; tail -4 gpurun_out/r06_job15/gputest.txt
timeout 600 python tests/probes/r06_players.py 256 > gpurun_out/r06_job15/players.txt 2>&1; cat gpurun_out/r06_job15/players.txt
python bench.py > gpurun_out/r06_job15/bench_default.json 2> gpurun_out/r06_job15/bench_default.err; tail -c 300 gpurun_out/r06_job15/bench_default.json
timeout 900 python tests/probes/fuzz_long_r6.py 400 "10 players" > gpurun_out/r06_job15/fuzz_p10.txt 2>&1; tail -4 gpurun_out/r06_job15/fuzz_p10.txt | cut -c1-300
