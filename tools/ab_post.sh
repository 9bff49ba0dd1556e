#!/bin/bash
# variant build for post-matcher experiments: recompiles ONLY k_post.hip.  usage: tools/ab_post.sh <tag> [extra hipcc flags]
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/ros_gpu_stereo_processor_amd/lib/obj; OUT=$ROOT/tools/ab; mkdir -p $OUT
hipcc -c --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value "$@" -I $ROOT/include -I $ROOT/ros_gpu_stereo_processor_amd/csrc $ROOT/ros_gpu_stereo_processor_amd/csrc/k_post.hip -o $OUT/post_$TAG.o
hipcc -shared --offload-arch=gfx950 -o $OUT/lib_$TAG.so $OUT/post_$TAG.o $(ls $OBJ/*.o | grep -v k_post.o)
echo built $OUT/lib_$TAG.so
