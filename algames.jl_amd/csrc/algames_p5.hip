// algames_p5.hip -- DoubleIntegrator (d = 2), Unicycle and Bicycle games with five players (n = 20: outside the single 16 x 16
// tile, dense Newton direction), base and extended ingredient sets: explicit instantiations for ALG_CFGS_P5.
#include "algames_kernels.hpp"

ALG_CFGS_P5(ALG_DEFINE_KERNELS)
