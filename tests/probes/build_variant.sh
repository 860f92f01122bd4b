#!/bin/bash
# usage: [UNITS="base_2 ext_di ..."] build_variant.sh NAME [extra hipcc flags...]  -> algames.jl_amd/lib/variants/NAME.so (select with ALGAMES_HIP_LIB=...)
# The listed translation units are rebuilt with the extra flags (default: every tile-path unit -- the nine base configurations, the EXT
# families and the team kernels); everything else (host code, dense-direction units) is linked from the default build
# (algames.jl_amd/lib/obj: python -c "import __graft_entry__ as g; g.build()" first).  UNITS="base_2" rebuilds the C2 kernels only (~25 s).
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/../.." && pwd); C=$R/algames.jl_amd/csrc; O=$R/algames.jl_amd/lib/variants/obj_$NAME; D=$R/algames.jl_amd/lib/obj
mkdir -p $O
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-invalid-offsetof -Wno-null-dereference -mllvm -disable-machine-licm --offload-compress $*"
# the tile-path units of the default build have the SI load/store optimizer off (__graft_entry__.HIP_FLAGS_TILE); LSO=1 keeps it on
if [ "${LSO:-0}" = 0 ]; then FL="$FL -Xclang -target-feature -Xclang -load-store-opt"; fi
FL="$FL -mllvm -amdgpu-inline-max-bb=100000"
UNITS=${UNITS:-"base_0 base_1 base_2 base_3 base_4 base_5 base_6 base_7 base_8 ext_di ext_uni ext_bic ext_di3 mw"}
J=0
for u in $UNITS; do
  case $u in
    base_*) /opt/rocm/bin/hipcc $FL -DALG_BASE_SEL=${u#base_} -c $C/algames_base.hip -o $O/algames_$u.hip.o & ;;
    *)      /opt/rocm/bin/hipcc $FL -c $C/algames_$u.hip -o $O/algames_$u.hip.o & ;;
  esac
  J=$((J+1)); if [ $J -ge 8 ]; then wait -n; J=$((J-1)); fi
done
wait
OBJS=""
for f in $D/*.hip.o; do b=$(basename $f); if [ -f $O/$b ]; then OBJS="$OBJS $O/$b"; else OBJS="$OBJS $f"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/algames.jl_amd/lib/variants/$NAME.so $OBJS
echo built $R/algames.jl_amd/lib/variants/$NAME.so
