// swg_reg_kernel instantiations (dp_reg.hpp); one translation unit per kernel family (aim_amd/build.py compiles them in parallel)
#define AIM_TU_SWG_REG
#include "dp_reg.hpp"
namespace aim { }
