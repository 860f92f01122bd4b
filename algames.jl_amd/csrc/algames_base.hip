// algames_base.hip -- kernels of the base configurations (ALG_CFGS_BASE of algames_kernels.hpp; the BASELINE shapes C1..C5 run
// these).  One translation unit per entry: the build compiles this file once per index with -DALG_BASE_SEL=<0..8>
// (__graft_entry__.HIP_UNITS), so the nine instantiations build in parallel and a change of one kernel shape rebuilds in seconds.
// Launched from algames_hip.hip, which declares them `extern template`.
#include "algames_kernels.hpp"

#ifndef ALG_BASE_SEL
#error "compile with -DALG_BASE_SEL=<index into ALG_CFGS_BASE>"
#endif
#if ALG_BASE_SEL == 0
ALG_DEFINE_KERNELS(ALG_MODEL_DOUBLE_INTEGRATOR, 1, 2, 0)
#elif ALG_BASE_SEL == 1
ALG_DEFINE_KERNELS(ALG_MODEL_DOUBLE_INTEGRATOR, 2, 2, 0)
#elif ALG_BASE_SEL == 2
ALG_DEFINE_KERNELS(ALG_MODEL_DOUBLE_INTEGRATOR, 3, 2, 0)
ALG_INSTANTIATE_HO_PARK(template, ALG_MODEL_DOUBLE_INTEGRATOR, 3, 2, 0, 4)          // (budgeted solve of the straggler hand-off, ALG_CFGS_HANDOFF)
#elif ALG_BASE_SEL == 3
ALG_DEFINE_KERNELS(ALG_MODEL_DOUBLE_INTEGRATOR, 4, 2, 0)
#elif ALG_BASE_SEL == 4
ALG_DEFINE_KERNELS(ALG_MODEL_DOUBLE_INTEGRATOR, 2, 3, 0)
#elif ALG_BASE_SEL == 5
ALG_DEFINE_KERNELS(ALG_MODEL_UNICYCLE, 1, 2, 0)
#elif ALG_BASE_SEL == 6
ALG_DEFINE_KERNELS(ALG_MODEL_UNICYCLE, 2, 2, 0)
#elif ALG_BASE_SEL == 7
ALG_DEFINE_KERNELS(ALG_MODEL_UNICYCLE, 3, 2, 0)
ALG_INSTANTIATE_HO_PARK(template, ALG_MODEL_UNICYCLE, 3, 2, 0, 4)
#elif ALG_BASE_SEL == 8
ALG_DEFINE_KERNELS(ALG_MODEL_UNICYCLE, 4, 2, 0)
ALG_INSTANTIATE_HO_PARK(template, ALG_MODEL_UNICYCLE, 4, 2, 0, 4)
#else
#error "ALG_BASE_SEL out of range (ALG_CFGS_BASE has nine entries)"
#endif
