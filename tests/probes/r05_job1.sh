#!/bin/bash
# Round-5 opening measurements on one box: baseline bench line, phase cycles of a lone wavefront vs four per SIMD, lane utilisation
#  of the step-wise kernels and of the three sweeps of k_direction (DIR_STOP variants).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_job1; mkdir -p $O; cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err
for g in 64 256 1024 4096; do ALGAMES_HIP_LIB=$R/tests/probes/lib_prof.so python tests/probes/phase_prof.py C2 $g 1 > $O/phase_c2_$g.txt 2>&1; done
bash tests/probes/r05_lane_util.sh C2 4096 $O
for v in ds0 ds1 ds1nogj ds2 ds3; do bash tests/probes/r05_lane_util.sh C2 4096 $O $R/algames.jl_amd/lib/variants/$v.so; done
bash tests/probes/r05_lane_util.sh C3 1024 $O
bash tests/probes/r05_lane_util.sh C5 64 $O
ls -la $O
# split value recursion (ALG_SPLITF) against the default: C2-family parity tests, then alternating benches
if [ -f $R/algames.jl_amd/lib/variants/splitf.so ]; then
  ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/splitf.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_batch.py tests/test_gpu_refinement.py -m gpu -x -q > $O/splitf_tests.txt 2>&1
  tail -5 $O/splitf_tests.txt
  bash tests/probes/ab.sh "--steps 20 --warmup 5" splitf > $O/ab_splitf_c2.txt 2>&1; cat $O/ab_splitf_c2.txt
  ALGAMES_HIP_LIB=$R/tests/probes/lib_prof.so true
fi
