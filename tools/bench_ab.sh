#!/bin/bash
# Same-box A/B of the pipelined bench for two library builds (GIP_LIB_PATH), alternating, N rounds.  usage: tools/bench_ab.sh <out> <libA> <libB> <rounds> [bench args...]
OUT=gpurun_out/$1; A=$2; B=$3; N=$4; shift 4
: > $OUT
for i in $(seq 1 $N); do
  for L in $A $B; do
    GIP_LIB_PATH=$L python3 bench.py --no-e2e --no-latency --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-40s value %9.1f ms/step %.4f kernel_ms %.4f alone %.4f' % ('$L'.split('/')[-1], d['value'], d['ms_per_step'], r['kernel_ms'], r['kernel_ms_alone']))" >> $OUT || exit 1
  done
done
cat $OUT
