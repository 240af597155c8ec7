// tu_dp_reg.hip -- the translation unit that instantiates the kernels of dp_reg.hpp (aim_amd/build.py compiles the tu_*.hip files in
// parallel and links them with aim_capi.hip into libaim_hip.so).
#define AIM_TU_DP_REG 1
#include "dp_reg.hpp"
