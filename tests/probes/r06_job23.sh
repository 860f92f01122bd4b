#!/bin/bash
# round 6, job 23: more sub-chains for the row-broadcast sums of the double-integrator kernels (3 / 4 instead of 2) and the forward sweep's w_k chain as
# two half-chains: same-box A/B at C2 (variants rebuild the C2 unit only)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job23; O=gpurun_out/r06_job23
bash tests/probes/ab.sh "--steps 20 --warmup 8" ch3 ch4 ws ch4ws 2>&1 | tee $O/ab_chains_c2.txt
for v in ch3 ch4 ws ch4ws; do
  ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/$v.so timeout 600 python -m pytest tests/test_gpu_full_batch.py tests/test_gpu_parity.py -q -x -k "C2 or c2 or case2 or golden" 2>&1 | tail -1 | sed "s/^/$v: /" | tee -a $O/parity_variants.txt
done
