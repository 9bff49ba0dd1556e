#!/bin/bash
# variant build for front-end experiments: recompiles ONLY k_image.hip.  usage: tools/ab_img.sh <tag> [extra hipcc flags]
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/ros_gpu_stereo_processor_amd/lib/obj; OUT=$ROOT/tools/ab; mkdir -p $OUT
hipcc -c --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value "$@" -I $ROOT/include -I $ROOT/ros_gpu_stereo_processor_amd/csrc $ROOT/ros_gpu_stereo_processor_amd/csrc/k_image.hip -o $OUT/img_$TAG.o
hipcc -shared --offload-arch=gfx950 -o $OUT/lib_$TAG.so $OUT/img_$TAG.o $(ls $OBJ/*.o | grep -v k_image.o)
echo built $OUT/lib_$TAG.so
