// tu_genasm.hip -- the translation unit that instantiates the kernels of genasm_wave.hpp (aim_amd/build.py compiles the tu_*.hip files in
// parallel and links them with aim_capi.hip into libaim_hip.so).
#define AIM_TU_GENASM 1
#include "genasm_wave.hpp"
