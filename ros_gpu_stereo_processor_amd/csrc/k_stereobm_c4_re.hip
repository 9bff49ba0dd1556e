// k_stereobm_c4_re.hip -- instantiations of the column-sum matcher (k_stereobm_c4_impl.h) for block radius 11, 12
#include "k_stereobm_c4_impl.h"

GIP_C4_RANGE_TU(gip_bm_c4_launch_re, GIP_C4_CASE(11) GIP_C4_CASE(12))
