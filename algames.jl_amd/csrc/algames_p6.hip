// algames_p6.hip -- DoubleIntegrator (d = 2), Unicycle and Bicycle games with six players (n = 24: outside the single 16 x 16
// tile, dense Newton direction), base and extended ingredient sets: explicit instantiations for ALG_CFGS_P6.
#include "algames_kernels.hpp"

ALG_CFGS_P6(ALG_DEFINE_KERNELS)
