// algames_qt.hip -- the quad-team translation unit: four games per 256-thread workgroup (algames_qt.hpp).
#define ALG_QT 1
#include "algames_qt.hpp"
#include "algames_qt_launch.h"

using namespace alg;

// newton_solve! of four games per workgroup (wavefront w: game 4 blockIdx + w); a wavefront whose game is done keeps serving the
// collective Newton directions of the others
template <class C>
__global__ void __launch_bounds__(256, C::WPE) k_newton_solve_qt(Params pr_arg, int init, uint64_t game_id0) {
    __shared__ QtBlock<C> blk;
    CPR pr = kernel_params();
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int g = (int)blockIdx.x * 4 + wv;
    Game G = game_view(pr, g);
    newton_solve<C>(pr, G, blk.per[wv], init, game_id0 + (uint64_t)g);
    qt_drain<C>(pr, blk.per[wv]);
}

using CfgC2 = Cfg<ALG_MODEL_DOUBLE_INTEGRATOR, 3, 2, 0, 1, 4>;

bool alg_qt_supported(const Params& p) {
    return p.model == ALG_MODEL_DOUBLE_INTEGRATOR && p.p == 3 && p.d == 2 && p.ext == 0 && p.B % 4 == 0 && p.B >= 4 &&
           p.kscratch_len >= (p.N - 1) * 96;
}
void alg_qt_launch_newton_solve(const Params& p, hipStream_t stream, int init, uint64_t game_id0) {
    hipLaunchKernelGGL((k_newton_solve_qt<CfgC2>), dim3(p.B / 4), dim3(256), 0, stream, p, init, game_id0);
}
