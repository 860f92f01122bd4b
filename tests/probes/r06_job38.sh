#!/bin/bash
# round 6, job 38: machine-scheduler strategies on the C2 unit (-mllvm -amdgpu-sched-strategy=...; scheduling only: bit-identical results expected)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job38; O=gpurun_out/r06_job38
for v in sch_max-ilp sch_max-memory-clause sch_gcn-iterative-ilp; do echo $v; python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/$v.so 2>&1 | grep "C2"; done | tee $O/bitwise_sched.txt
bash tests/probes/ab.sh "--steps 20 --warmup 8" sch_max-ilp sch_max-memory-clause sch_gcn-iterative-ilp 2>&1 | tee $O/ab_sched_c2.txt
