#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_job11; mkdir -p $O
bash tests/probes/ab.sh "--steps 20 --warmup 8" r6_nogate r6_nobook r6_nofwdp r6_none3 r6_ref3 > $O/ab_micro_c2.txt 2>&1; cat $O/ab_micro_c2.txt
