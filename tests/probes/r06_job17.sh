#!/bin/bash
# round 6, job 17: grouped line search on the one-wavefront double-integrator kernels (ALG_LSM_DI1W) -- bitwise tests, the ten-player hard seed in the
# arbiter tests, same-box A/B against the variant without it (homogeneous C2 must not move; perturbed batches with and without the hand-off)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job17; O=gpurun_out/r06_job17
timeout 900 python -m pytest tests/test_gpu_line_search_batch.py tests/test_gpu_handoff.py tests/test_gpu_fuzz.py -q -x -k "line_search or handoff or hard_seeds or arbiter_on" 2>&1 | tail -5 | tee $O/tests.txt
bash tests/probes/ab.sh "--steps 20 --warmup 8" nolsm 2>&1 | tee $O/ab_lsm_di1w_c2.txt
for args in "--perturb 0.3" "--perturb 0.3 --handoff 16" "--perturb 0.15" "--config C4 --perturb 0.3"; do
  echo "== $args" | tee -a $O/ab_lsm_di1w_hetero.txt
  for round in 1 2; do for v in default nolsm; do
    if [ $v = default ]; then unset ALGAMES_HIP_LIB; else export ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/$v.so; fi
    python bench.py --no-cpu-baseline --no-pmc --steps 5 --warmup 3 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$v', '%.4g game-iterations/s' % d['value'], '%.3f ms' % d['ms_per_step'], 'mean/max %.3f' % c['iters_per_game_mean_over_max_rank0'], 'converged', d['games_converged'], 'iters', c['newton_iters_per_solve_total'])" | tee -a $O/ab_lsm_di1w_hetero.txt
  done; done
done
unset ALGAMES_HIP_LIB
