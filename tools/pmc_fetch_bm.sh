#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the matcher for library variants (tools/ab/lib_<tag>.so), separate passes.  usage: tools/pmc_fetch_bm.sh out tag...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift
: > $OUT
for tag in "$@"; do for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/pf_${tag}_$c; rm -rf $d
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -o r -- python3 $R/tools/bm_ab.py $R/tools/ab/lib_$tag.so > /dev/null 2>&1
  echo "== $tag" >> $OUT; python3 $R/tools/pmc_summary.py $(find $d -name "*counter_collection.csv" | head -1) k_bm_c4 >> $OUT
done; done
cat $OUT
