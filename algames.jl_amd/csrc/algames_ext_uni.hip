// algames_ext_uni.hip -- kernels of the extended ingredient set (state bounds, walls, circles): explicit
// instantiations of the entry points of algames_kernels.hpp for ALG_CFGS_EXT_UNI.  Launched from algames_hip.hip.
#include "algames_kernels.hpp"

ALG_CFGS_EXT_UNI(ALG_DEFINE_KERNELS)
