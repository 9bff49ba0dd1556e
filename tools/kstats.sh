#!/bin/bash
# per-kernel stats of a bench.py run: tools/kstats.sh <outfile-in-gpurun_out> [bench args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift
rm -rf /tmp/ks_run
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_run -o r -- python3 $R/bench.py --no-cpu-baseline --no-latency --no-e2e "$@" > ${OUT%.csv}.bench.txt 2>&1
cp $(find /tmp/ks_run -name "*kernel_stats.csv" | head -1) $OUT
python3 - "$OUT" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %5s avg_us %9.1f  pct %5s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
