#!/bin/bash
# split value recursion (default build) against the round-4 form (variant nosplit): tests, A/B, phase cycles
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_job2; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_batch.py tests/test_gpu_refinement.py tests/test_gpu_parity_ext.py -m gpu -x -q > $O/splitf_tests.txt 2>&1
tail -15 $O/splitf_tests.txt
bash tests/probes/ab.sh "--steps 20 --warmup 5" nosplit > $O/ab_splitf_c2.txt 2>&1; cat $O/ab_splitf_c2.txt
for g in 256 4096; do ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/prof.so python tests/probes/phase_prof.py C2 $g 1 > $O/phase_c2_$g.txt 2>&1; cat $O/phase_c2_$g.txt; done
bash tests/probes/ab.sh "--steps 20 --warmup 5 --games-per-gpu 8192" nosplit > $O/ab_splitf_c4.txt 2>&1; cat $O/ab_splitf_c4.txt
