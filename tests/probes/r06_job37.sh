#!/bin/bash
# round 6, job 37: GPU suite on the final sources (new: chunk shapes of the fused pass, the long run's outliers, seed 101156), smoke(), the default bench lines
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job37; O=gpurun_out/r06_job37
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^arbiter consulted\|^forward error\|^decision\|^status differs\|^device omega\|^seed " | tail -12 > $O/gputest_final.txt; tail -3 $O/gputest_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > $O/bench_steps20_warmup5.json 2>/dev/null; tail -c 300 $O/bench_steps20_warmup5.json
