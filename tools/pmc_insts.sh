#!/bin/bash
# VALU instruction count (and VALU-active quad-cycles) of the matcher for library variants tools/ab/lib_<tag>.so, one rocprofv3 pass each.
# usage (through gpurun, from the repo root): bash tools/pmc_insts.sh <outfile> tag1 [tag2 ...]   (AB_* env selects the shape)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift
: > $OUT
for tag in "$@"; do
  d=/tmp/pmci_$tag; rm -rf $d
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $d -o r -- python3 $R/tools/bm_ab.py $R/tools/ab/lib_$tag.so > /tmp/pmci_run.log 2>&1 || { echo "failed: $tag" >> $OUT; continue; }
  echo "== $tag" >> $OUT
  python3 $R/tools/pmc_summary.py $(find $d -name "*counter_collection.csv" | head -1) k_bm_c4 >> $OUT
done
cat $OUT
