// tu_dp_strip.hip -- the translation unit that instantiates the kernels of dp_strip.hpp (aim_amd/build.py compiles the tu_*.hip files in
// parallel and links them with aim_capi.hip into libaim_hip.so).
#define AIM_TU_DP_STRIP 1
#include "dp_strip.hpp"
