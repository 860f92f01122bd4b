// algames_p7.hip -- DoubleIntegrator (d = 2), Unicycle and Bicycle games with seven players (dense Newton direction), base and extended
// ingredient sets: explicit instantiations for ALG_CFGS_P7.
#include "algames_kernels.hpp"

ALG_CFGS_P7(ALG_DEFINE_KERNELS)
