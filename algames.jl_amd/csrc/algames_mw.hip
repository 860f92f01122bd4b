// Explicit instantiations of the team kernels (several wavefronts per game; ALG_CFGS_MW of algames_kernels.hpp): their own
// translation unit so that they compile in parallel with the others.
#include "algames_kernels.hpp"
ALG_CFGS_MW(ALG_DEFINE_MW)
// ... and the kernels that resume the games a budgeted one-wavefront solve parked (straggler hand-off, alg_set_handoff)
ALG_CFGS_HANDOFF(ALG_DEFINE_HO_RESUME)
