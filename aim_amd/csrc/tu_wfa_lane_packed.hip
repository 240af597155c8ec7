// tu_wfa_lane_packed.hip -- the translation unit that instantiates the kernels of wfa_lane_packed.hpp (aim_amd/build.py compiles the tu_*.hip files in
// parallel and links them with aim_capi.hip into libaim_hip.so).
#define AIM_TU_WFA_LANE_PACKED 1
#include "wfa_lane_packed.hpp"
