#!/bin/bash
# round 6, job 5: per-chunk phase A (bitwise + A/B + traffic), rollout with staged controls, dense refinement while it contracts (fuzz seeds of round 5, Q2 / Q4), suite
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_job5; mkdir -p $O
echo "bitwise default vs r6_nopch:" > $O/bitwise.txt; python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/r6_nopch.so 2>&1 | tail -6 >> $O/bitwise.txt; cat $O/bitwise.txt
bash tests/probes/ab.sh "--steps 20 --warmup 8" r6_nopch > $O/ab_pch_c2.txt 2>&1; cat $O/ab_pch_c2.txt
bash tests/probes/ab.sh "--config C5 --games-per-gpu 4096 --steps 10 --warmup 4" r6_nopch > $O/ab_pch_c5_4096.txt 2>&1; cat $O/ab_pch_c5_4096.txt
bash tests/probes/ab.sh "--config C3 --games-per-gpu 4096 --steps 10 --warmup 4" r6_nopch > $O/ab_pch_c3_4096.txt 2>&1; cat $O/ab_pch_c3_4096.txt
timeout 1200 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_refinement.py -q -s 2>&1 | grep -v "^decision" | tail -25 > $O/tests_fuzz.txt; cat $O/tests_fuzz.txt | cut -c1-400
( echo "config games waves value corrections"
  for spec in "Q2 4096" "Q4 1024" "C3 1024" "C2 512" "C5 1024"; do set -- $spec
    python bench.py --config $1 --games-per-gpu $2 --steps 10 --warmup 4 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(c['name'], c['games_per_gpu'], c['wavefronts_per_game'], '%.4g' % d['value'], 'corrections', c['direction_refinement']['correction_solves_rank0'])"
  done
  python bench.py --config C5 --mpc-steps 200 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5 loop 64 x 200', '%.4g' % d['value'], '%.1f ms' % d['ms_per_step'])"
) > $O/other_shapes.txt 2>&1; cat $O/other_shapes.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/gputest.txt; tail -6 $O/gputest.txt | cut -c1-300
