#!/bin/bash
# round 5, after the grouped line search reached the one-wavefront unicycle kernels: GPU suite, C5 loops (64 seeds team of four; 4096 seeds one wavefront), bench lines
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v "^arbiter consulted\|^forward error\|^decision" | tail -12 > gpurun_out/r05_gputest_final.txt; tail -3 gpurun_out/r05_gputest_final.txt
for m in 0 1; do
ALGAMES_LS_MULTI=$m python bench.py --config C5 --mpc-steps 50 --games-per-gpu 4096 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 loop 4096 seeds x 50 steps, one wavefront per game, ALGAMES_LS_MULTI=$m:', '%.4g' % j['value'], 'game-iterations/s', '%.1f ms' % j['ms_per_step'])"
ALGAMES_LS_MULTI=$m python bench.py --config C5 --mpc-steps 200 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 loop 64 seeds x 200 steps, team of four, ALGAMES_LS_MULTI=$m:', '%.4g' % j['value'], 'game-iterations/s', '%.1f ms' % j['ms_per_step'])"
done 2>&1 | tee gpurun_out/r05_ab_lsmulti_final.txt
python bench.py > gpurun_out/bench_r05_default.json 2> gpurun_out/bench_r05_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > gpurun_out/bench_r05_steps20_warmup5.json 2>/dev/null
( echo "config games waves value"
  for spec in "C2 16384" "C2 512" "C3 4096" "C5 1024" "C5 4096" "Q2 4096" "Q4 1024"; do set -- $spec
    python bench.py --config $1 --games-per-gpu $2 --steps 10 --warmup 4 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(c['name'], c['games_per_gpu'], c['wavefronts_per_game'], '%.4g' % d['value'], 'corrections', c['direction_refinement']['correction_solves_rank0'])"
  done ) > gpurun_out/r05_other_shapes.txt 2>&1
python tests/probes/hetero.py 2>&1 | grep -v histogram > gpurun_out/r05_hetero.txt
cat gpurun_out/r05_other_shapes.txt; grep "spread\|games in one" gpurun_out/r05_hetero.txt | cut -c1-220
