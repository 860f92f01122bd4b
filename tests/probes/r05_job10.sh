#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_ab
bash tests/probes/ab.sh "--config Q2 --steps 10 --warmup 3" hiponly dnolso 2>&1 | tee gpurun_out/r05_ab/ab_dense_q2.txt
bash tests/probes/ab.sh "--config Q4 --steps 4 --warmup 1" hiponly dnolso 2>&1 | tee gpurun_out/r05_ab/ab_dense_q4.txt
ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/dnolso.so python bench.py --config Q2 --steps 5 --warmup 2 --no-cpu-baseline --no-pmc | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['direction_refinement'], d['config']['newton_iters_per_solve_total'])"
ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/dnolso.so timeout 900 python -m pytest tests/test_gpu_parity_quad.py tests/test_gpu_parity_dense.py tests/test_gpu_refinement.py "tests/test_gpu_fuzz.py::test_direction_backward_error_against_the_arbiter" -m gpu -q -x 2>&1 | tail -5
