#!/bin/bash
# registers / spills of ONE kernel instantiation with the build's flags: kres.sh 'k_mpc_loop<Cfg<ALG_MODEL_UNICYCLE, 3, 2, 0, 4>>(Params, int, uint64_t, double*)' [extra flags]
R=$(cd "$(dirname "$0")/../.." && pwd); K=$1; shift; mkdir -p /tmp/kres; T=/tmp/kres/k_$$
echo "#include \"$R/algames.jl_amd/csrc/algames_kernels.hpp\"
template __global__ void $K;" > $T.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-invalid-offsetof -mllvm -disable-machine-licm -Xclang -target-feature -Xclang -load-store-opt -Wno-unused-command-line-argument --cuda-device-only -S -o $T.s $T.hip "$@" && grep -E "^\s+\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):" $T.s | tr -s ' \n' ' '; echo; cp $T.s /tmp/kres/last.s; rm -f $T.hip $T.s
