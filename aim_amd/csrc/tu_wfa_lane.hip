// tu_wfa_lane.hip -- the translation unit that instantiates the kernels of wfa_lane.hpp (aim_amd/build.py compiles the tu_*.hip files in
// parallel and links them with aim_capi.hip into libaim_hip.so).
#define AIM_TU_WFA_LANE 1
#include "wfa_lane.hpp"
