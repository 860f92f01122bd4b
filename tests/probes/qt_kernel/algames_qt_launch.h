// host-side entry points of the quad-team translation unit (algames_qt.hip)
#pragma once
#include "algames_device.hpp"
bool alg_qt_supported(const alg::Params& p);
void alg_qt_launch_newton_solve(const alg::Params& p, hipStream_t stream, int init, uint64_t game_id0);
