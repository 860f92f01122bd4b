#!/bin/bash
# round 6, job 6: the round-5 long-run seeds with full output, DoubleIntegrator d = 1 parity, bitwise of the direction's split into phase functions
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_job6; mkdir -p $O
timeout 900 python -m pytest "tests/test_gpu_fuzz.py::test_fuzz_long_run_seeds_of_round_5" -q -s 2>&1 | grep -v "^decision" | grep "Error\|^E  \|passed\|failed\|arbiter\|status differs" | cut -c1-700 > $O/seeds.txt; cat $O/seeds.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "1-1-7 or 2-1-9 or 3-1-6 or 4-1-8 or d1 or case16 or case17 or case18 or case19" 2>&1 | tail -15 > $O/d1.txt; cat $O/d1.txt | cut -c1-300
echo "bitwise default (direction split into phase functions) vs the library of job 5:" > $O/bitwise_split.txt; python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/r6_ref2.so 2>&1 | tail -6 >> $O/bitwise_split.txt; cat $O/bitwise_split.txt
bash tests/probes/ab.sh "--steps 20 --warmup 8" r6_ref2 > $O/ab_split_c2.txt 2>&1; cat $O/ab_split_c2.txt
