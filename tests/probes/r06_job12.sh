#!/bin/bash
# round 6, job 12: row-broadcast FMA chains as two interleaved half-chains: A/B on the BASELINE shapes, parity of the variant
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_job12; mkdir -p $O
bash tests/probes/ab.sh "--steps 20 --warmup 8" r6_rdsplit > $O/ab_rdsplit_c2.txt 2>&1; cat $O/ab_rdsplit_c2.txt
bash tests/probes/ab.sh "--config C3 --steps 10 --warmup 4" r6_rdsplit > $O/ab_rdsplit_c3.txt 2>&1; cat $O/ab_rdsplit_c3.txt
bash tests/probes/ab.sh "--config C5 --mpc-steps 100 --steps 3 --warmup 1" r6_rdsplit > $O/ab_rdsplit_c5loop.txt 2>&1; cat $O/ab_rdsplit_c5loop.txt
bash tests/probes/ab.sh "--games-per-gpu 512 --steps 20 --warmup 8" r6_rdsplit > $O/ab_rdsplit_c2_512.txt 2>&1; cat $O/ab_rdsplit_c2_512.txt
ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/r6_rdsplit.so timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_batch.py tests/test_gpu_fuzz.py tests/test_gpu_refinement.py tests/test_gpu_line_search_batch.py -q 2>&1 | tail -8 > $O/tests_variant.txt; cat $O/tests_variant.txt | cut -c1-300
