#!/bin/bash
# Fast variant build for matcher experiments: recompiles ONLY k_stereobm_c4_ra.hip (block radius 2..4: the KITTI shape) with the given flags and links it with the other
# objects of the normal build (ros_gpu_stereo_processor_amd/lib/obj).  usage: tools/ab_c4.sh <tag> [extra hipcc flags]
set -e
TAG=$1; shift
TU=${C4_TU:-ra}   # translation unit to recompile: ra (R 2-4), rb (5-6), rc (7-8), rd (9-10), re (11-12), rf (13-14), rg (15)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/ros_gpu_stereo_processor_amd/lib/obj; OUT=$ROOT/tools/ab; mkdir -p $OUT
hipcc -c --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-function "$@" -I $ROOT/include -I $ROOT/ros_gpu_stereo_processor_amd/csrc \
  $ROOT/ros_gpu_stereo_processor_amd/csrc/k_stereobm_c4_$TU.hip -o $OUT/c4_$TAG.o
hipcc -shared --offload-arch=gfx950 -o $OUT/lib_$TAG.so $OUT/c4_$TAG.o $(ls $OBJ/*.o | grep -v k_stereobm_c4_$TU.o)
echo built $OUT/lib_$TAG.so
