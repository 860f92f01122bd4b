#!/bin/bash
# round 6, job 8: dense refinement (stall rule, eight corrections): seeds, the whole fuzz / refinement / dense files, Q2 / Q4 rates; then the suite
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_job8; mkdir -p $O
timeout 600 python tests/probes/r06_seed_solve.py 400051 > $O/seed_solve_400051.txt 2>&1; grep "default" $O/seed_solve_400051.txt | cut -c1-300
timeout 1200 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_refinement.py tests/test_gpu_parity_quad.py tests/test_gpu_parity_dense.py -q -s 2>&1 | grep -v "^decision" | grep "Error\|^E  \|passed\|failed\|arbiter consulted\|status differs" | cut -c1-600 > $O/tests_dense.txt; tail -12 $O/tests_dense.txt
( echo "config games waves value corrections"
  for spec in "Q2 4096" "Q4 1024"; do set -- $spec
    python bench.py --config $1 --games-per-gpu $2 --steps 10 --warmup 4 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(c['name'], c['games_per_gpu'], c['wavefronts_per_game'], '%.4g' % d['value'], 'corrections', c['direction_refinement']['correction_solves_rank0'], 'iters', c['newton_iters_per_solve_total'])"
  done ) > $O/quad_shapes.txt 2>&1; cat $O/quad_shapes.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/gputest.txt; tail -6 $O/gputest.txt | cut -c1-300
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.json
