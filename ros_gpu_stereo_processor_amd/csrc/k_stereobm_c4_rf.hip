// k_stereobm_c4_rf.hip -- instantiations of the column-sum matcher (k_stereobm_c4_impl.h) for block radius 13, 14
#include "k_stereobm_c4_impl.h"

GIP_C4_RANGE_TU(gip_bm_c4_launch_rf, GIP_C4_CASE(13) GIP_C4_CASE(14))
