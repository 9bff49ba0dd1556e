// k_stereobm_c4_rb.hip -- instantiations of the column-sum matcher (k_stereobm_c4_impl.h) for block radius 5, 6
#include "k_stereobm_c4_impl.h"

GIP_C4_RANGE_TU(gip_bm_c4_launch_rb, GIP_C4_CASE(5) GIP_C4_CASE(6))
