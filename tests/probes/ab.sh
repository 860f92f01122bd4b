#!/bin/bash
# usage (GPU box): ab.sh "bench args" variant1 variant2 ...   -- alternates the default library and the variants, 3 rounds
R=$GRAFT_REPO_ROOT; ARGS=$1; shift
for round in 1 2 3; do
  for v in default "$@"; do
    if [ $v = default ]; then unset ALGAMES_HIP_LIB; else export ALGAMES_HIP_LIB=$R/algames.jl_amd/lib/variants/$v.so; fi
    val=$(python $R/bench.py --no-cpu-baseline --no-pmc $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g %.3f ms' % (d['value'], d['roofline']['kernel_ms_avg']))")
    echo "$v: $val"
  done
done
