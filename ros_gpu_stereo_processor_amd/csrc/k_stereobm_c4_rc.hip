// k_stereobm_c4_rc.hip -- instantiations of the column-sum matcher (k_stereobm_c4_impl.h) for block radius 7, 8
#include "k_stereobm_c4_impl.h"

GIP_C4_RANGE_TU(gip_bm_c4_launch_rc, GIP_C4_CASE(7) GIP_C4_CASE(8))
