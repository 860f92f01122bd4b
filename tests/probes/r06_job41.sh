#!/bin/bash
# round 6, job 41: is the automatic kernel shape still the fastest after the round's changes to the one-wavefront kernels?  (waves per game 1 / 2 / 4 against the default)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job41; O=gpurun_out/r06_job41
run() { python bench.py "$@" --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(c['name'], c['games_per_gpu'], 'waves', c['wavefronts_per_game'], '%.4g game-iterations/s' % d['value'], '%.3f ms' % d['ms_per_step'])"; }
( for w in 0 1 2 4; do run --config C3 --steps 10 --warmup 4 --waves-per-game $w; done
  for g in 512 1024 2048; do for w in 0 1 2 4; do run --config C2 --games-per-gpu $g --steps 10 --warmup 4 --waves-per-game $w; done; done
  for g in 2048; do for w in 0 1 2; do run --config C3 --games-per-gpu $g --steps 10 --warmup 4 --waves-per-game $w; done; done
  for w in 0 1 2 4; do run --config C5 --mpc-steps 200 --steps 2 --warmup 1 --waves-per-game $w; done ) 2>&1 | tee $O/shapes.txt
