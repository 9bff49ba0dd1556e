#!/usr/bin/env python3
"""Copy the round's rocprofv3 summaries from gpurun_out/prof_r03 (written by tools/prof_round.sh on the GPU box) into profiles/ under the
names profiles/README.md lists, and derive profiles/traffic.json (HBM bytes per matcher launch) and profiles/r03_stretch.txt from them."""
import csv, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "prof_r03"), os.path.join(ROOT, "profiles")
TAG = "r03"


def copy(src, dst):
    s = os.path.join(SRC, src)
    if os.path.exists(s):
        shutil.copyfile(s, os.path.join(DST, dst))
        return True
    return False


for wl in ["", "hd1080_", "uhd4k_", "euroc_", "refdefault_"]:
    for d1 in ["", "depth1_"]:
        copy(f"{TAG}_{wl}{d1}kernel_stats.csv", f"{TAG}_{wl}{d1}kernel_stats.csv")
        copy(f"{TAG}_{wl}{d1}bench.json", f"{TAG}_{wl}{d1}bench_under_rocprof.json")
    copy(f"{TAG}_{wl}pmc_sq1_k_bm_c4.txt", f"{TAG}_{wl}pmc_sq1_k_bm_c4.txt")
    if wl:
        copy(f"{TAG}_{wl}pmc_fetch_size.txt", f"{TAG}_{wl}pmc_fetch_size.txt")
        copy(f"{TAG}_{wl}pmc_write_size.txt", f"{TAG}_{wl}pmc_write_size.txt")
for f in ["pmc_sq2_k_bm_c4", "pmc_sq_all_kernels", "pmc_fetch_size", "pmc_write_size", "rowsplit4_kernel_stats", "rowsplit4_serial"]:
    copy(f"{TAG}_{f}.txt", f"{TAG}_{f}.txt")
copy(f"{TAG}_rowsplit_bench.json", f"{TAG}_rowsplit_bench.json")


def counter_of(path, kernel, counter):
    """value of `counter` (per launch average) in a tools/pmc_summary.py file for the first kernel whose name contains `kernel`"""
    cur = None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
        elif cur and kernel in cur:
            m = re.match(r"\s+(\S+)\s+([0-9.eE+]+)", line)
            if m and m.group(1) == counter:
                return float(m.group(2))
    return None


NOTE = ("rocprofv3 FETCH_SIZE reads exactly half of the streamed bytes on gfx950 at every access width (profiles/r02_pmc_calibration.txt: "
        "1 GiB copy kernels, 1/2/4/8/16 B per lane), WRITE_SIZE is exact; traffic = 2 x FETCH_SIZE + WRITE_SIZE of the dominant kernel, separate --pmc passes "
        "(r03_[<workload>_]pmc_fetch_size.txt, r03_[<workload>_]pmc_write_size.txt)")
traffic = {"note": NOTE, "workloads": {}}
for key, wl in (("kitti", ""), ("hd1080", "hd1080_"), ("uhd4k", "uhd4k_"), ("euroc", "euroc_"), ("refdefault", "refdefault_")):
    bpath = os.path.join(DST, f"{TAG}_{wl}bench_under_rocprof.json")
    fpath, wpath = os.path.join(DST, f"{TAG}_{wl}pmc_fetch_size.txt"), os.path.join(DST, f"{TAG}_{wl}pmc_write_size.txt")
    if not (os.path.exists(bpath) and os.path.exists(fpath) and os.path.exists(wpath)):
        continue
    bench = json.loads(open(bpath).read().strip().splitlines()[-1])
    kern = bench["roofline"]["kernel"]
    kname = kern.split("<")[0]
    fetch_kb, write_kb = counter_of(fpath, kname, "FETCH_SIZE"), counter_of(wpath, kname, "WRITE_SIZE")
    if fetch_kb is None or write_kb is None:
        continue
    traffic["workloads"][key] = {"batch": bench["config"]["pairs_per_step_per_gpu"], "kernel": kern, "fetch_size_counter_bytes": fetch_kb * 1024.0,
                                 "write_size_bytes": write_kb * 1024.0, "fetch_correction": 2.0, "hbm_bytes_per_launch": 2.0 * fetch_kb * 1024.0 + write_kb * 1024.0}
    print("traffic.json: %s %.1f MB per launch" % (key, traffic["workloads"][key]["hbm_bytes_per_launch"] / 1e6))
json.dump(traffic, open(os.path.join(DST, "traffic.json"), "w"), indent=1)


def stats(path):
    out = {}
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "")
        if n.startswith("k_"):
            out[n] = float(r["AverageNs"]) / 1e3
    return out


alone, cos = stats(os.path.join(DST, f"{TAG}_depth1_kernel_stats.csv")), stats(os.path.join(DST, f"{TAG}_kernel_stats.csv"))
with open(os.path.join(DST, f"{TAG}_stretch.txt"), "w") as f:
    f.write("# per-kernel average duration (us), KITTI workload, 64 pairs per launch: alone (bench.py --depth 1) vs co-scheduled (default: 3 batches in flight,\n"
            "# matchers on one shared stream), from r03_depth1_kernel_stats.csv / r03_kernel_stats.csv; stretch = co-scheduled / alone\n")
    f.write("%-52s %10s %12s %8s\n" % ("kernel", "alone_us", "cosched_us", "stretch"))
    for n in sorted(alone, key=lambda k: -alone[k]):
        if n in cos:
            f.write("%-52s %10.1f %12.1f %8.2f\n" % (n[:52], alone[n], cos[n], cos[n] / alone[n]))
    f.write("# round 1 (32-row speckle strips, 57 KB LDS per workgroup): k_spk_strip 29.7 alone, 772 co-scheduled = 26x (LDS starvation next to the matcher's\n"
            "# 20 KB workgroups); 8-row strips (10 KB) removed most of it.  The remaining 2-3x of the small kernels is time-sharing of the VALU with the matcher.\n")
print(open(os.path.join(DST, f"{TAG}_stretch.txt")).read())
