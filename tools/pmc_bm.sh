#!/bin/bash
# SQ counters of the matcher for one or more library builds (tools/ab/lib_<tag>.so), each in its own rocprofv3 passes.
# usage (through gpurun, from the repo root): bash tools/pmc_bm.sh <outfile> <kernel-filter> tag1 [tag2 ...]   (AB_* env selects the shape)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; FILT=$2; shift 2
: > $OUT
for tag in "$@"; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
             "SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_LDS_DATA_FIFO_FULL"; do
    i=$((i+1)); d=/tmp/pmc_${tag}_$i; rm -rf $d
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o r -- python3 $R/tools/bm_ab.py $R/tools/ab/lib_$tag.so > /tmp/pmc_run.log 2>&1 || { echo "pmc run failed: $tag set $i" >> $OUT; tail -3 /tmp/pmc_run.log >> $OUT; continue; }
    f=$(find $d -name "*counter_collection.csv" | head -1)
    echo "== $tag (set $i)" >> $OUT
    python3 $R/tools/pmc_summary.py $f "$FILT" >> $OUT
  done
done
tail -80 $OUT
