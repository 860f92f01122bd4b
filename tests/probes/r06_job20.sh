#!/bin/bash
# round 6, job 20: the row-broadcast half-chains once more (variant rdsplit = every tile-path unit with -DALG_R6_ROWDOT_SPLIT=1): which assertions of the
# suite move, with their numbers; same-box A/B at C2
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job20; O=gpurun_out/r06_job20
ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/rdsplit.so timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^arbiter consulted\|^forward error\|^decision\|^status differs\|^device omega" > $O/tests_variant_full.txt
grep -E "^E  |FAILED|passed|failed" $O/tests_variant_full.txt | cut -c1-600 | tail -40
bash tests/probes/ab.sh "--steps 20 --warmup 8" rdsplit 2>&1 | tee $O/ab_rdsplit_c2.txt
