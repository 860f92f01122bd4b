// algames_ext_bic.hip -- kernels of the extended ingredient set (state bounds, walls, circles; bicycle model): explicit
// instantiations of the entry points of algames_kernels.hpp for ALG_CFGS_EXT_BIC.  Launched from algames_hip.hip.
#include "algames_kernels.hpp"

ALG_CFGS_EXT_BIC(ALG_DEFINE_KERNELS)
