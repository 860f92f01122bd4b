// Explicit instantiations of the DoubleIntegratorGame(d = 1) kernels (ALG_CFGS_DI1 of algames_kernels.hpp): their own translation unit so
// that they compile in parallel with the others.
#include "algames_kernels.hpp"
ALG_CFGS_DI1(ALG_DEFINE_KERNELS)
