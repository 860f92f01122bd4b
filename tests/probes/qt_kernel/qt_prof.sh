#!/bin/bash
# Per-phase / per-role cycle profile of the quad-team Newton direction.  build (CPU container): bash tests/probes/qt_prof.sh build ;
# on the GPU box: bash tests/probes/qt_prof.sh run [games]
set -e
R=$(cd $(dirname $0)/../.. && pwd); D=$R/algames.jl_amd/lib/obj
if [ "$1" = build ]; then
  FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-invalid-offsetof -mllvm -disable-machine-licm -DALG_PHASE_PROF"
  for f in algames_hip algames_qt; do /opt/rocm/bin/hipcc $FL -c $R/algames.jl_amd/csrc/$f.hip -o /tmp/qtprof_$f.o & done; wait
  O=""; for f in algames_ext_di algames_ext_uni algames_ext_bic algames_ext_di3 algames_mw algames_quad algames_quad_ext algames_di3 algames_mw_dense algames_p5 algames_p6; do O="$O $D/$f.hip.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tests/probes/lib_qtprof.so /tmp/qtprof_algames_hip.o /tmp/qtprof_algames_qt.o $O
  echo built
else
  shift; ALGAMES_HIP_LIB=$R/tests/probes/lib_qtprof.so python $R/tests/probes/qt_prof.py "$@"
fi
