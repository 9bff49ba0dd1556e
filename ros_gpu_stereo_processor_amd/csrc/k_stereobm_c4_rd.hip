// k_stereobm_c4_rd.hip -- instantiations of the column-sum matcher (k_stereobm_c4_impl.h) for block radius 9, 10
#include "k_stereobm_c4_impl.h"

GIP_C4_RANGE_TU(gip_bm_c4_launch_rd, GIP_C4_CASE(9) GIP_C4_CASE(10))
