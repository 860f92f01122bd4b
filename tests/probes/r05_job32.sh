#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r05_lsm
timeout 900 python -m pytest tests/test_gpu_line_search_batch.py -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/r05_lsm/test.txt
for i in 1 2; do for m in 0 1; do
ALGAMES_LS_MULTI=$m python bench.py --config C5 --mpc-steps 200 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 loop ls_multi=$m:', '%.4g' % j['value'], j['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r05_lsm/ab_c5loop.txt
