// tu_dp_wave.hip -- the translation unit that instantiates the kernels of dp_wave.hpp (aim_amd/build.py compiles the tu_*.hip files in
// parallel and links them with aim_capi.hip into libaim_hip.so).
#define AIM_TU_DP_WAVE 1
#include "dp_wave.hpp"
