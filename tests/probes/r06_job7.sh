#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_job7; mkdir -p $O
for s in 400051 400040; do timeout 600 python tests/probes/r06_seed_solve.py $s > $O/seed_solve_$s.txt 2>&1; cat $O/seed_solve_$s.txt | cut -c1-400; done
