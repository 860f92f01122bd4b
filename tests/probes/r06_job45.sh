#!/bin/bash
# round 6, job 45: prefetch depth of the forward / costate sweeps on the final C2 kernel (default 4; variants sd2 / sd6): bit-identity, same-box A/B
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r06_job45; O=gpurun_out/r06_job45
for v in sd2 sd6; do echo $v; python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/$v.so 2>&1 | grep C2; done | tee $O/bitwise_sd.txt
bash tests/probes/ab.sh "--steps 20 --warmup 8" sd2 sd6 2>&1 | tee $O/ab_sd_c2.txt
