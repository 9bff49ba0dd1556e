#!/bin/bash
# (pixel, disparity) pairs per second of gipk_stereobm (XSOBEL x2 + matcher + L/R check) over the whole fast-path grid, one batch alone.
# usage (through gpurun): bash tools/matcher_grid.sh <outfile>
OUT=gpurun_out/$1; L=ros_gpu_stereo_processor_amd/lib/libgpuimageproc_hip.so
echo "# gipk_stereobm, 32 pairs of 1242x375 per call, ms per call and T (px,d)/s over the valid output area; every cell runs k_bm_c4 (no fallback kernel)" > $OUT
echo "# ndisp  block: 5        9        13       15       19       21       25       31" >> $OUT
for nd in 16 32 48 64 80 96 112 128 144 160 176 192 208 224 240 256; do
  line=$(printf "%-6s" $nd)
  for b in 5 9 13 15 19 21 25 31; do
    ms=$(AB_ND=$nd AB_BLOCK=$b AB_B=32 python3 tools/bm_ab.py $L 2>/dev/null | grep median | sed 's/.*median \([0-9.]*\) ms.*/\1/')
    rate=$(python3 -c "w=1242-$nd+1-($b-1); h=375-($b-1); print('%.2f' % (32*w*h*$nd/($ms*1e-3)/1e12))")
    line="$line  $ms/$rate"
  done
  echo "$line" >> $OUT
done
cat $OUT
