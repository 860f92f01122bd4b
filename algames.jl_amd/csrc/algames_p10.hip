// algames_p10.hip -- DoubleIntegrator (d = 2), Unicycle and Bicycle games with ten players, the reference's cap (options.jl:68): dense Newton
// direction in the TIGHT LDS layout (DirLds<C, true>), base and extended ingredient sets: explicit instantiations for ALG_CFGS_P10.
#include "algames_kernels.hpp"

ALG_CFGS_P10(ALG_DEFINE_KERNELS)
