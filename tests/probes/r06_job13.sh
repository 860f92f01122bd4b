#!/bin/bash
# round 6, job 13: resume kernels without the line search's LDS staging (four teams per CU): hand-off tests + budget sweep; d = 1 fuzz
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_job13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_handoff.py "tests/test_gpu_fuzz.py::test_fuzz_double_integrator_d1" -q 2>&1 | tail -6 > $O/tests.txt; cat $O/tests.txt | cut -c1-300
timeout 600 python tests/probes/r06_handoff.py C2 4096 0.3 > $O/handoff_c2.txt 2>&1; cat $O/handoff_c2.txt
timeout 600 python tests/probes/r06_handoff.py C5 4096 0.3 > $O/handoff_c5.txt 2>&1; cat $O/handoff_c5.txt
