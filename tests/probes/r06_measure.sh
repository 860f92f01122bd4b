#!/bin/bash
# Round-6 evidence run on the GPU box (gpurun): kernel traces + PMC passes of the four BASELINE shapes (tests/probes/prof_r02.sh, 10 steps after
# 8 warm-up launches), the six-pass counter summary of C2, other shapes, strong-scaling shares, heterogeneous batches (bench.py --perturb with and
# without the hand-off), the default bench line.  Summaries: tests/probes/mk_r06_evidence.py.
R=$GRAFT_REPO_ROOT
export PROF_STEPS=10 PROF_WARMUP=8
bash $R/tests/probes/prof_r02.sh r06_c2
bash $R/tests/probes/prof_r02.sh r06_c4 --config C4
bash $R/tests/probes/prof_r02.sh r06_c3 --config C3
PROF_STEPS=3 PROF_WARMUP=1 bash $R/tests/probes/prof_r02.sh r06_c5mpc --config C5 --mpc-steps 200
cd $R
for g in 4096 2048 1024 512; do
  python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-pmc --games-per-gpu $g > gpurun_out/r06_share_$g.json 2>/dev/null
done
( echo "config games waves value"
  for spec in "C2 16384" "C2 512" "C3 4096" "C5 1024" "C5 4096" "Q2 4096" "Q4 1024"; do set -- $spec
    python bench.py --config $1 --games-per-gpu $2 --steps 10 --warmup 4 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(c['name'], c['games_per_gpu'], c['wavefronts_per_game'], '%.4g' % d['value'], 'corrections', c['direction_refinement']['correction_solves_rank0'])"
  done
  python bench.py --config C5 --mpc-steps 50 --games-per-gpu 4096 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5 loop 4096 seeds x 50 steps, one wavefront per game:', '%.4g' % d['value'])"
) > gpurun_out/r06_other_shapes.txt 2>&1
( for spec in "C2 0.3 0" "C2 0.3 16" "C5 0.3 0" "C5 0.3 14" "C3 0.2 0" "C3 0.2 16"; do set -- $spec
    python bench.py --config $1 --games-per-gpu 4096 --perturb $2 --handoff $3 --steps 5 --warmup 3 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(c['name'], 'perturb', c['perturb'], 'hand-off', c['handoff'], '%.4g game-iterations/s' % d['value'], '%.3f ms' % d['ms_per_step'], 'mean/max iterations %.2f' % c['iters_per_game_mean_over_max_rank0'], 'converged', d['games_converged'])"
  done ) > gpurun_out/r06_hetero.txt 2>&1
python bench.py > gpurun_out/bench_r06_default.json 2> gpurun_out/bench_r06_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > gpurun_out/bench_r06_steps20_warmup5.json 2>/dev/null
tail -c 400 gpurun_out/bench_r06_default.json
bash $R/tests/probes/r05_pmc.sh gpurun_out/r06_pmc_final > /dev/null 2>&1
find gpurun_out/r06_* -name "*agent_info*" -delete 2>/dev/null
cat gpurun_out/r06_other_shapes.txt gpurun_out/r06_hetero.txt
