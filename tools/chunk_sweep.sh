#!/bin/bash
# Ground truth for the matcher's row-chunk chooser (launch_c4): time gipk_stereobm with the chunk count forced to 1..N through the
# GIP_C4_CHUNKS knob (the library under test must be a measurement build: tools/ab_build.sh WORKTREE knobs "-DC4_MEASURE_KNOBS"; the product build ignores the variable), then with the chooser's own pick.  usage (through gpurun): AB_* env selects the shape; tools/chunk_sweep.sh <outfile> n1 n2 ...
OUT=gpurun_out/$1; shift
echo "shape W=${AB_W:-1242} H=${AB_H:-375} B=${AB_B:-64} ND=${AB_ND:-128} BLOCK=${AB_BLOCK:-9}" >> $OUT
L=ros_gpu_stereo_processor_amd/lib/libgpuimageproc_hip.so
for n in "$@"; do
  printf "  chunks %-3s " $n >> $OUT
  if [ "$n" = auto ]; then python3 tools/bm_ab.py $L 2>&1 | grep median >> $OUT; else GIP_C4_CHUNKS=$n python3 tools/bm_ab.py $L 2>&1 | grep median >> $OUT; fi
done
