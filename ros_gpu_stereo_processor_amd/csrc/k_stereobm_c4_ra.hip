// k_stereobm_c4_ra.hip -- instantiations of the column-sum matcher (k_stereobm_c4_impl.h) for block radius 2, 3, 4
#include "k_stereobm_c4_impl.h"

GIP_C4_RANGE_TU(gip_bm_c4_launch_ra, GIP_C4_CASE(2) GIP_C4_CASE(3) GIP_C4_CASE(4))
