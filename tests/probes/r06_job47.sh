#!/bin/bash
# round 6, job 47: the final library (sweep depth 6 / 4): the whole GPU suite, smoke, bench lines of the four BASELINE shapes
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job47; O=gpurun_out/r06_job47
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 | tee $O/gputest_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
run() { python bench.py "$@" --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(c['name'], c['games_per_gpu'], 'waves', c['wavefronts_per_game'], '%.4g game-iterations/s' % d['value'], '%.3f ms' % d['ms_per_step'], 'frac %.4f' % d['roofline']['frac'])"; }
( run --steps 20 --warmup 5; run --config C4 --steps 10 --warmup 4; run --config C3 --steps 20 --warmup 8; run --config C5 --mpc-steps 200 --steps 3 --warmup 1
  run --config C3 --games-per-gpu 4096 --steps 10 --warmup 4; run --config C5 --games-per-gpu 4096 --steps 10 --warmup 4; run --config C5 --games-per-gpu 1024 --steps 10 --warmup 4 ) | tee $O/rates_final.txt
