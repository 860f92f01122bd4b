#!/bin/bash
# last evidence job of round 5: the whole GPU suite on the final binary, the C5 loop under rocprofv3 + counters, the default bench line
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v "^arbiter consulted\|^forward error\|^decision" | tail -12 > gpurun_out/r05_gputest_final.txt; cat gpurun_out/r05_gputest_final.txt | tail -3
PROF_STEPS=3 PROF_WARMUP=1 bash $R/tests/probes/prof_r02.sh r05_c5mpc --config C5 --mpc-steps 200 > /dev/null 2>&1
cd $R
python bench.py > gpurun_out/bench_r05_default.json 2> gpurun_out/bench_r05_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > gpurun_out/bench_r05_steps20_warmup5.json 2>/dev/null
( echo "config games waves value"
  for spec in "C2 16384" "C2 512" "C3 4096" "C5 1024" "C5 4096" "Q2 4096" "Q4 1024"; do set -- $spec
    python bench.py --config $1 --games-per-gpu $2 --steps 10 --warmup 4 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(c['name'], c['games_per_gpu'], c['wavefronts_per_game'], '%.4g' % d['value'], 'corrections', c['direction_refinement']['correction_solves_rank0'])"
  done ) > gpurun_out/r05_other_shapes.txt 2>&1
find gpurun_out/r05_* -name "*agent_info*" -delete 2>/dev/null
cat gpurun_out/r05_other_shapes.txt; tail -c 300 gpurun_out/r05_c5mpc/bench.json
