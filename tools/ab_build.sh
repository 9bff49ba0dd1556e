#!/bin/bash
# Build libgpuimageproc_hip.so from a given git revision (or WORKTREE for the current files) into tools/ab/lib_<tag>.so
# so that two variants can be timed in ONE process on ONE device (tools/bm_ab.py).
set -e
REV=$1; TAG=${2:-$1}; EXTRA="$3"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/tools/ab; mkdir -p $OUT
TMP=$(mktemp -d)
if [ "$REV" = "WORKTREE" ]; then
  cp -r $ROOT/ros_gpu_stereo_processor_amd/csrc $TMP/csrc; cp -r $ROOT/include $TMP/include
else
  mkdir -p $TMP/csrc $TMP/include
  git -C $ROOT archive $REV ros_gpu_stereo_processor_amd/csrc include | tar -x -C $TMP
  mv $TMP/ros_gpu_stereo_processor_amd/csrc/* $TMP/csrc/
fi
SRCS=$(ls $TMP/csrc/*.hip $TMP/csrc/*.cpp)
OBJS=""
for f in $SRCS; do
  o=$TMP/$(basename $f).o
  X=""; case $f in *.cpp) X="-x hip";; esac
  hipcc -c --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value $EXTRA -I $TMP/include -I $TMP/csrc $X $f -o $o &
  OBJS="$OBJS $o"
done
wait
hipcc -shared --offload-arch=gfx950 -o $OUT/lib_$TAG.so $OBJS
rm -rf $TMP
echo built $OUT/lib_$TAG.so
