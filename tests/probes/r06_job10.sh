#!/bin/bash
# round 6, job 10: lane-0 bookkeeping in one round trip, gate pass four items in flight, forward-sweep prologue: bitwise + A/B against the library of job 9; new tests; suite
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_job10; mkdir -p $O
echo "bitwise default vs the library of job 9:" > $O/bitwise.txt; python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/r6_ref3.so 2>&1 | tail -6 >> $O/bitwise.txt; cat $O/bitwise.txt
bash tests/probes/ab.sh "--steps 20 --warmup 8" r6_ref3 > $O/ab_c2.txt 2>&1; cat $O/ab_c2.txt
bash tests/probes/ab.sh "--config C3 --steps 10 --warmup 4" r6_ref3 > $O/ab_c3.txt 2>&1; cat $O/ab_c3.txt
bash tests/probes/ab.sh "--config C5 --mpc-steps 100 --steps 3 --warmup 1" r6_ref3 > $O/ab_c5loop.txt 2>&1; cat $O/ab_c5loop.txt
timeout 900 python -m pytest tests/test_gpu_refinement.py tests/test_gpu_parity_ext.py -q -k "failed_correction or unsupported or t_elap" 2>&1 | tail -8 > $O/tests_new.txt; cat $O/tests_new.txt | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/gputest.txt; tail -6 $O/gputest.txt | cut -c1-300
