// algames_quad_ext.hip -- QuadrotorGame kernels with the extended ingredient set (state bounds, walls, circles and the 3-D half:
// spherical collision avoidance, Wall3D, Cylinder on pz[i][1:3]): explicit instantiations for ALG_CFGS_QUAD_EXT.
#include "algames_kernels.hpp"

ALG_CFGS_QUAD_EXT(ALG_DEFINE_KERNELS)
