#!/bin/bash
# builds tests/probes/lib_prof.so here (CPU container) with -DALG_PHASE_PROF: bash tests/probes/phase_prof.sh build ; on the GPU box: bash tests/probes/phase_prof.sh run C3 1024
set -e
R=$(cd $(dirname $0)/../.. && pwd)
if [ "$1" = build ]; then
  FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-invalid-offsetof -mllvm -disable-machine-licm -mllvm -amdgpu-inline-max-bb=100000 -DALG_PHASE_PROF"
  for f in algames_hip algames_ext_di algames_ext_uni algames_ext_bic algames_ext_di3 algames_mw algames_quad algames_quad_ext algames_di3 algames_mw_dense algames_p5 algames_p6; do /opt/rocm/bin/hipcc $FL -c $R/algames.jl_amd/csrc/$f.hip -o /tmp/prof_$f.o & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tests/probes/lib_prof.so /tmp/prof_algames_hip.o /tmp/prof_algames_ext_di.o /tmp/prof_algames_ext_uni.o /tmp/prof_algames_ext_bic.o /tmp/prof_algames_ext_di3.o /tmp/prof_algames_mw.o /tmp/prof_algames_quad.o /tmp/prof_algames_quad_ext.o /tmp/prof_algames_di3.o /tmp/prof_algames_mw_dense.o /tmp/prof_algames_p5.o /tmp/prof_algames_p6.o
  echo built
else
  shift; ALGAMES_HIP_LIB=$R/tests/probes/lib_prof.so python $R/tests/probes/phase_prof.py "$@"
fi
