// algames_quad.hip -- kernels of the QuadrotorGame instantiations (Cfg::DENSE: dense per-player Jacobian blocks from forward-mode
// differentiation of the RK2 step, LDS-resident dense Newton direction with tiled f64 MFMA products): explicit instantiations
// of the entry points of algames_kernels.hpp for ALG_CFGS_QUAD.  Launched from algames_hip.hip.
#include "algames_kernels.hpp"

ALG_CFGS_QUAD(ALG_DEFINE_KERNELS)
