// wfa_lane.hpp -- short-read WFA fast path (one pair per lane).  Placeholder until the
// kernel lands: nothing is routed here yet.
#pragma once

#include "aim_device.hpp"

namespace aim {

inline bool wfa_lane_supported(const aim_params_t &) { return false; }
inline void wfa_lane_plan(const aim_params_t &, uint32_t, uint32_t *grid, uint32_t *block, size_t *lds)
{
    *grid = 8; *block = kWave; *lds = 0;
}
inline void wfa_lane_launch(const aim_params_t &, uint32_t, uint32_t, size_t, const KArgs &, hipStream_t) {}

}  // namespace aim
