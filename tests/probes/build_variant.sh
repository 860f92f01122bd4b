#!/bin/bash
# usage: build_variant.sh NAME [extra hipcc flags...]  -> algames.jl_amd/lib/variants/NAME.so (select with ALGAMES_HIP_LIB=...)
# The tile-path translation units are rebuilt with the extra flags; the dense-direction units are linked from the default build
# (algames.jl_amd/lib/obj, python -c "import __graft_entry__ as g; g.build()" first).
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/../.." && pwd); C=$R/algames.jl_amd/csrc; O=$R/algames.jl_amd/lib/variants/obj_$NAME; D=$R/algames.jl_amd/lib/obj
mkdir -p $O
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-invalid-offsetof -mllvm -disable-machine-licm $*"
for f in algames_hip algames_ext_di algames_ext_uni algames_ext_bic algames_ext_di3 algames_mw; do
  /opt/rocm/bin/hipcc $FL -c $C/$f.hip -o $O/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/algames.jl_amd/lib/variants/$NAME.so $O/*.o $D/algames_quad.hip.o $D/algames_quad_ext.hip.o $D/algames_di3.hip.o $D/algames_mw_dense.hip.o $D/algames_p5.hip.o $D/algames_p6.hip.o
echo built $R/algames.jl_amd/lib/variants/$NAME.so
