// dp_group_kernel instantiations (dp_group.hpp)
#define AIM_TU_DP_GROUP
#include "dp_group.hpp"
