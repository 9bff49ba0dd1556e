#!/bin/bash
# What happens between two consecutive matcher launches of the pipelined bench: rocprofv3 kernel trace -> gap (start[i+1] - end[i]) per matcher,
# and the kernels that start inside each gap.  usage (through gpurun): bash tools/gap_probe.sh [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/p_gap
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_gap -o r -- python3 $R/bench.py --no-latency --no-e2e --no-cpu-baseline "$@" > /dev/null 2>&1
python3 - <<'PY' > $R/gpurun_out/gap_probe.txt
import csv,glob
f=glob.glob('/tmp/p_gap/**/*kernel_trace.csv',recursive=True)[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'],r.get('Queue_Id','?')) for r in csv.DictReader(open(f))]
rows.sort()
m=[r for r in rows if 'k_bm_c4' in r[2]]
gaps=[]
for a,b in zip(m[-21:-1],m[-20:]):
    gaps.append((b[0]-a[1])/1e3)
print("matcher queue ids:", sorted(set(r[3] for r in m)))
print("gaps between the last 20 matcher launches (us):", ["%.1f"%g for g in gaps])
print("mean gap %.1f us, mean matcher %.1f us"%(sum(gaps)/len(gaps), sum((r[1]-r[0]) for r in m[-20:])/20e3))
a,b=m[-10],m[-9]
print("kernels active or starting inside one gap [%d..%d]:"%(0,(b[0]-a[1])//1000))
for r in rows:
    if r[1]>a[1]-20000 and r[0]<b[0]+20000 and 'k_bm_c4' not in r[2]:
        print("  %-40s q%s start %+8.1f us end %+8.1f us (relative to matcher end)"%(r[2].split('(')[0][-40:], r[3], (r[0]-a[1])/1e3,(r[1]-a[1])/1e3))
PY
cat $R/gpurun_out/gap_probe.txt
