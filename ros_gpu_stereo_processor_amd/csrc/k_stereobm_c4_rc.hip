// k_stereobm_c4_rc.hip -- instantiations of the column-sum matcher (k_stereobm_c4_impl.h) for block radius 9, 10, 11
#include "k_stereobm_c4_impl.h"

GIP_C4_RANGE_TU(gip_bm_c4_launch_rc, GIP_C4_CASE(9) GIP_C4_CASE(10) GIP_C4_CASE(11))
