// algames_p9.hip -- DoubleIntegrator (d = 2), Unicycle and Bicycle games with nine players (dense Newton direction), base and extended
// ingredient sets: explicit instantiations for ALG_CFGS_P9.
#include "algames_kernels.hpp"

ALG_CFGS_P9(ALG_DEFINE_KERNELS)
