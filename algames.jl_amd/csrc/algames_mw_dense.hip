// Explicit instantiations of the team kernels (four wavefronts per game) of the dense-direction configurations
// (ALG_CFGS_MW_DENSE of algames_kernels.hpp): QuadrotorGame p = 2..4 and DoubleIntegrator d = 3 with p = 3, 4.
#include "algames_kernels.hpp"
ALG_CFGS_MW_DENSE(ALG_DEFINE_MW)
