// algames_ext_di3.hip -- kernels of the extended ingredient set with 3-D positions (DoubleIntegrator d = 3: spherical
// collision avoidance, Wall3D, Cylinder on top of state bounds / walls / circles): explicit instantiations of the entry
// points of algames_kernels.hpp for ALG_CFGS_EXT_DI3.  Launched from algames_hip.hip.
#include "algames_kernels.hpp"

ALG_CFGS_EXT_DI3(ALG_DEFINE_KERNELS)
