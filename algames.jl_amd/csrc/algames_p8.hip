// algames_p8.hip -- DoubleIntegrator (d = 2), Unicycle and Bicycle games with eight players (dense Newton direction), base and extended
// ingredient sets: explicit instantiations for ALG_CFGS_P8.
#include "algames_kernels.hpp"

ALG_CFGS_P8(ALG_DEFINE_KERNELS)
