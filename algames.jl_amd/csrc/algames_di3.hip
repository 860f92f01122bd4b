// algames_di3.hip -- DoubleIntegrator in three dimensions with p = 1, 3, 4 players (n = 6, 18, 24: outside the single 16 x 16 tile
// of the structured Newton direction, so these take the dense variant), base and extended ingredient set: explicit
// instantiations for ALG_CFGS_DI3D.
#include "algames_kernels.hpp"

ALG_CFGS_DI3D(ALG_DEFINE_KERNELS)
