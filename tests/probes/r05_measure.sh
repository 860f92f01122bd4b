#!/bin/bash
# Round-5 evidence run on the GPU box (gpurun): kernel traces + PMC passes of the four BASELINE shapes (tests/probes/prof_r02.sh, 10 steps after
# 8 warm-up launches so that the shader clock has settled), the other shapes, the strong-scaling shares and the default bench line.
# Summaries: tests/probes/mk_r05_evidence.py.
R=$GRAFT_REPO_ROOT
export PROF_STEPS=10 PROF_WARMUP=8
bash $R/tests/probes/prof_r02.sh r05_c2
bash $R/tests/probes/prof_r02.sh r05_c4 --config C4
bash $R/tests/probes/prof_r02.sh r05_c3 --config C3
PROF_STEPS=3 PROF_WARMUP=1 bash $R/tests/probes/prof_r02.sh r05_c5mpc --config C5 --mpc-steps 200
cd $R
for g in 4096 2048 1024 512; do
  python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-pmc --games-per-gpu $g > gpurun_out/r05_share_$g.json 2>/dev/null
done
( echo "config games waves value"
  for spec in "C2 16384" "C3 4096" "C5 1024" "Q2 4096" "Q4 1024"; do set -- $spec
    python bench.py --config $1 --games-per-gpu $2 --steps 10 --warmup 4 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(c['name'], c['games_per_gpu'], c['wavefronts_per_game'], '%.4g' % d['value'], 'corrections', c['direction_refinement']['correction_solves_rank0'])"
  done ) > gpurun_out/r05_other_shapes.txt 2>&1
python bench.py > gpurun_out/bench_r05_default.json 2> gpurun_out/bench_r05_default.err
tail -c 400 gpurun_out/bench_r05_default.json
bash $R/tests/probes/r05_pmc.sh gpurun_out/r05_pmc_final > /dev/null 2>&1
find gpurun_out/r05_* -name "*agent_info*" -delete 2>/dev/null
du -sh gpurun_out/r05_c2 gpurun_out/r05_c3 | tail -3
