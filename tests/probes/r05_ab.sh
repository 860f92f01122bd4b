#!/bin/bash
# usage (GPU box): r05_ab.sh TAG "bench args" variant...   alternating benches of the default library and variants + bitwise comparison
R=$GRAFT_REPO_ROOT; TAG=$1; ARGS=$2; shift; shift; O=$R/gpurun_out/r05_ab; mkdir -p $O; cd $R
bash tests/probes/ab.sh "$ARGS" "$@" > $O/ab_$TAG.txt 2>&1; cat $O/ab_$TAG.txt
for v in "$@"; do echo "bitwise default vs $v:"; python tests/probes/bitwise_ab.py algames.jl_amd/lib/libalgames_hip.so algames.jl_amd/lib/variants/$v.so 2>&1 | tail -6; done | tee $O/bitwise_$TAG.txt
