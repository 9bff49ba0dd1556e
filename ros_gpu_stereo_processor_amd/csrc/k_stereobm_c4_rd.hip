// k_stereobm_c4_rd.hip -- instantiations of the column-sum matcher (k_stereobm_c4_impl.h) for block radius 12, 13, 14, 15
#include "k_stereobm_c4_impl.h"

GIP_C4_RANGE_TU(gip_bm_c4_launch_rd, GIP_C4_CASE(12) GIP_C4_CASE(13) GIP_C4_CASE(14) GIP_C4_CASE(15))
