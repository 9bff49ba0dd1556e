// k_stereobm_c4_rg.hip -- instantiations of the column-sum matcher (k_stereobm_c4_impl.h) for block radius 15
#include "k_stereobm_c4_impl.h"

GIP_C4_RANGE_TU(gip_bm_c4_launch_rg, GIP_C4_CASE(15))
