cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/p_cmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_cmp -o r -- python3 $R/bench.py --no-latency --no-e2e --no-cpu-baseline > $R/gpurun_out/cmp_bench.json 2> /dev/null
grep k_bm_c4 $(find /tmp/p_cmp -name "*kernel_stats.csv" | head -1) | cut -d, -f1-8 > $R/gpurun_out/cmp_stats.txt
python3 - <<'PY' >> $R/gpurun_out/cmp_stats.txt
import csv,glob
f=glob.glob('/tmp/p_cmp/**/*kernel_trace.csv',recursive=True)[0]
d=[ (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6 for r in csv.DictReader(open(f)) if 'k_bm_c4' in r['Kernel_Name']]
print("launch durations ms:", ["%.3f"%x for x in d])
print("last 20 mean %.4f"%(sum(d[-20:])/20))
PY
