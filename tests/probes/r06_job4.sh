#!/bin/bash
# round 6, job 4: the tightened bounds (gate row scale, golden 1e-8, boundary ulps), hand-off + rank tests, dense-gap probe, phase profile, then the whole suite
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_job4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_handoff.py tests/test_gpu_refinement.py tests/test_gpu_boundary.py "tests/test_gpu_parity.py::test_golden_solutions_gpu" tests/test_gpu_bench_ranks.py -q -s 2>&1 | grep -v "^arbiter consulted\|^decision" | tail -40 > $O/tests_new.txt; cat $O/tests_new.txt | cut -c1-300
for s in 400051 400040; do timeout 600 python tests/probes/r06_dense_gap.py $s > $O/dense_gap_$s.txt 2>&1; done; head -60 $O/dense_gap_400051.txt | cut -c1-330
ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/prof.so python tests/probes/phase_prof.py C2 4096 > $O/phase_cycles_c2_4096.txt 2>&1; cat $O/phase_cycles_c2_4096.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/gputest.txt; tail -6 $O/gputest.txt | cut -c1-300
