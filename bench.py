#!/usr/bin/env python3
"""Throughput benchmark of the stereo hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the whole per-frame chain (rectify L/R + colour L -> StereoBM with L/R check -> speckle filter ->
float disparity + 32-byte PointCloud2 records) over one batch of synthetic stereo pairs that are already resident in HBM.
Workload = BASELINE.json configs[1]: 1242x375 mono8, ndisp 128, block 9.  Frames are sharded across ranks (independent
units, weak scaling): rank 0 builds the rectification tables on the host and broadcasts CameraInfo + tables once over
RCCL; there is no data-path collective.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H, NDISP, BLOCK = 1242, 375, 128, 9
ENCODING = "mono8"
# BASELINE.json configs[1] (the one `metric` is quoted on, default) and configs[2] (north_star's second size; --workload hd1080)
WORKLOADS = {
    # depth = batches in flight per GPU, shared = their matcher kernels go to ONE stream (both measured per workload)
    # batch: pairs resident per step.  Measured (one MI355X, gpurun_out/bsz*.log): kitti 54.1 k pairs/s at 32, 57.3 k at 64, 60.7 k at 128, 61.7 k at 256, 61.6 k at 512
    # (the matcher's last partly filled round of workgroups and the hand-over between two matcher launches are amortised over more pairs); refdefault
    # 69.4 / 72.5 / 73.6 k at 64 / 128 / 256; euroc and hd1080 do not move.  256 pairs are 8 GB per pipeline including the 3.8 GB of cloud records.
    "kitti": dict(W=1242, H=375, NDISP=128, BLOCK=9, ENCODING="mono8", batch=256, depth=3, shared=True, kernel="k_bm_c4<8,4,1>",
                  name="KITTI-size 1242x375 mono8, ndisp=128 block=9, full chain rectify->StereoBM(XSOBEL,tex10,uniq15,L/R disp12=1)"
                       "->speckle(800,5)->f32 disparity+PointCloud2"),
    "hd1080": dict(W=1920, H=1080, NDISP=256, BLOCK=21, ENCODING="bayer_rggb8", batch=8, depth=3, shared=False, kernel="k_bm_c4<4,10,4>",
                   name="1920x1080 bayer_rggb8, ndisp=256 block=21, full chain debayer(gray L/R + colour L)->rectify->StereoBM(XSOBEL,tex10,uniq15,"
                        "L/R disp12=1)->speckle(800,5)->f32 disparity+PointCloud2"),
    # (three frames in flight, each with its matcher on its own stream: 1.84 k frames/s; two frames with a shared matcher stream 1.56 k, with their
    # own 1.65 k -- since the L/R check of rows wider than 1536 pixels keeps its keys in 4 bytes per column and fits next to the matcher)
    "uhd4k": dict(W=4096, H=2160, NDISP=192, BLOCK=15, ENCODING="mono8", batch=1, depth=3, shared=False, kernel="k_bm_c4<6,7,2>",
                  name="4096x2160 mono8, ndisp=192 block=15, ONE frame per step on one GPU (BASELINE configs[4] without the row split), full chain "
                       "rectify->StereoBM(XSOBEL,tex10,uniq15,L/R disp12=1)->speckle(800,5)->f32 disparity+PointCloud2"),
    # BASELINE configs[0]'s size and matcher shape (the reference's own CPU-runnable case) on the GPU path
    # (four batches in flight with their matchers on their own streams: 90.2 k pairs/s against 87.4 k for three + a shared matcher stream -- this
    # shape's step is bound by the small kernels' chain, not by the matcher chain)
    "euroc": dict(W=752, H=480, NDISP=64, BLOCK=15, ENCODING="mono8", batch=64, depth=4, shared=False, kernel="k_bm_c4<4,7,1>",
                  name="EuRoC-size 752x480 mono8, ndisp=64 block=15, full chain rectify->StereoBM(XSOBEL,tex10,uniq15,L/R disp12=1)->speckle(800,5)->f32 disparity+PointCloud2"),
    # the reference constructor's matcher shape, createStereoBM(48, 19) (reference src/GPUStereoProcessor.cpp:18-20), at the KITTI size
    "refdefault": dict(W=1242, H=375, NDISP=48, BLOCK=19, ENCODING="mono8", batch=256, depth=3, shared=True, kernel="k_bm_c4<6,9,1,*,2> (two images of the batch per wave)",
                       name="1242x375 mono8, ndisp=48 block=19 (the reference's createStereoBM(48, 19)), full chain rectify->StereoBM(XSOBEL,tex10,uniq15,L/R disp12=1)"
                            "->speckle(800,5)->f32 disparity+PointCloud2"),
}
HBM_PEAK_GBS = 8000.0
# Algorithmic VALU floor of an exact SAD block matcher on this ISA, in issue slots (one VOP2/VOP3 wave-instruction = 1 slot per lane)
# per (pixel, disparity) -- DESIGN.md section 3.1: 2 row contributions (entering, leaving) at the SAD unit's 4 byte abs-diffs per slot
# = 0.5; vertical running sum 0.5 + horizontal sliding window 1.0 + winner-take-all min 0.5 as packed-u16 instructions (2 values each).
# It contains no index tracking, no uniqueness / texture test, no sub-pixel step and no halo work: a floor, not a target.
VALU_FLOOR_SLOTS = 2.5
VALU_PEAK_LANE_SLOTS = 39.3e12   # measured VOP3 issue rate, tools/ubench_valu.hip (profiles/r01_ubench_valu.txt): 4.2 cycles per wave-instruction per SIMD  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def bm_kwargs():
    # cv::StereoBM defaults (XSOBEL, cap 31, texture 10, uniqueness 15) + the L/R consistency check north_star asks for
    return dict(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=BLOCK, min_disparity=0, num_disparities=NDISP,
                texture_threshold=10, uniqueness_ratio=15, disp12_max_diff=DISP12, speckle_window_size=0, speckle_range=0)


DISP12 = 1      # --ablate lr (experiment) sets -1: no L/R consistency check


SPECKLE_SIZE, SPECKLE_DIFF = 800, 5.0  # cfg/GPU.cfg:34-35 defaults


def make_inputs(batch, seed0, nseeds=4):
    from ros_gpu_stereo_processor_amd import synth
    calib = synth.scaled_calibration(W, H)
    ls, rs = [], []
    for i in range(nseeds):
        l, r, _, _ = synth.raw_pair(W, H, NDISP, seed=seed0 + i, calib=calib)
        if ENCODING != "mono8":
            l, r = synth.bayer_mosaic(synth.colorize(l, 1), ENCODING), synth.bayer_mosaic(synth.colorize(r, 2), ENCODING)
        ls.append(l)
        rs.append(r)
    idx = np.arange(batch) % nseeds
    return np.stack(ls)[idx], np.stack(rs)[idx], calib


def cpu_chain(O, lr, rr, calib, maps, po):
    """The same chain on the host cores with the oracle (the restatement of the reference's CPU path)."""
    if ENCODING == "mono8":
        rl, rrect = O.remap(lr, *maps[0]), O.remap(rr, *maps[1])
        bgr = O.gray_to_bgr(rl)
    else:
        rl, rrect = O.remap(O.bayer_to_gray(lr, ENCODING), *maps[0]), O.remap(O.bayer_to_gray(rr, ENCODING), *maps[1])
        bgr = O.remap(O.bayer_to_bgr(lr, ENCODING), *maps[0])
    d16 = O.stereobm(rl, rrect, po)
    d16 = O.filter_speckles(d16, -16, SPECKLE_SIZE, int(SPECKLE_DIFF * 16))
    d32 = O.disparity_to_f32(d16, calib["left"]["P"][2], calib["right"]["P"][2])
    xyz = O.reproject_to_3d(d32, O.stereo_q(calib["left"]["P"], calib["right"]["P"]), True)
    return d16, O.pack_pointcloud2(xyz, bgr)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_chain_timed(workload, budget_s, max_pairs, seed0=100):
    """Times the oracle chain on the host cores; honours OMP_NUM_THREADS.  Returns (pairs, seconds, threads)."""
    global W, H, NDISP, BLOCK, ENCODING
    wl = WORKLOADS[workload]
    W, H, NDISP, BLOCK, ENCODING = wl["W"], wl["H"], wl["NDISP"], wl["BLOCK"], wl["ENCODING"]
    from oracle import oracle as O
    lraw, rraw, calib = make_inputs(2, seed0, nseeds=2)
    maps = [O.init_undistort_rectify_map(calib[s]["K"], calib[s]["D"], calib[s]["R"], calib[s]["P"], W, H) for s in ("left", "right")]
    po = O.bm_params(**bm_kwargs())
    cpu_chain(O, lraw[0], rraw[0], calib, maps, po)  # warm-up
    n, t0 = 0, time.perf_counter()
    while n < max_pairs and (time.perf_counter() - t0) < budget_s:
        cpu_chain(O, lraw[n % len(lraw)], rraw[n % len(rraw)], calib, maps, po)
        n += 1
    return n, time.perf_counter() - t0, O.num_threads()


def _cpu_share():
    """CPUs this process may really use: affinity mask capped by the cgroup quota (the GPU box shows 128 logical CPUs but a
    1-GPU job owns a 16-CPU share; 128 OpenMP threads on that share ran the chain SLOWER than one thread)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(workload, budget_s=5.0, workers=None):
    """Whole-host throughput of the oracle chain: one single-threaded worker process per available CPU, every worker running
    the chain on its own frames (frames are independent -- the same sharding the GPUs use), none of them touching the GPU.
    This is how a CPU box would be used for THROUGHPUT; OpenCV's own StereoBM only splits a 1242x375 frame into 5 row stripes."""
    import subprocess
    workers = workers or min(_cpu_share(), 64)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-chain-only", workload, str(budget_s)]
    procs = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(workers)]
    res = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=budget_s * 6 + 60)
            res.append(json.loads(out.strip().splitlines()[-1]))
        except Exception:
            p.kill()
    if not res:
        return dict(value=None, unit="stereo pairs/s", cores=0, kind="port", sample="CPU baseline workers failed")
    pairs = sum(r["pairs"] for r in res)
    secs = max(r["seconds"] for r in res)
    one = sum(r["pairs"] / r["seconds"] for r in res) / len(res)
    wl = WORKLOADS[workload]
    return dict(value=round(pairs / secs, 2), unit="stereo pairs/s", cores=len(res), kind="port",
                sample="%d pairs of the same %dx%d ndisp=%d block=%d chain (%sremap L/R, StereoBM+L/R check, speckle, reproject, PointCloud2 pack) "
                       "in %.1f s: %d single-threaded worker processes of the C oracle, one per CPU of this job's share, each on its own frames; "
                       "OpenCV itself is not installable here"
                       % (pairs, wl["W"], wl["H"], wl["NDISP"], wl["BLOCK"], "" if wl["ENCODING"] == "mono8" else "debayer, ", secs, len(res)),
                cpu_model=_cpu_model(), logical_cpus_visible=os.cpu_count(), per_worker_pairs_per_s=round(one, 3))


def single_pair_latency(lraw, rraw, calib, params, device, iters=30):
    """ms per frame when ONE pair goes through the reference's per-frame schedule (gip_process_pair = StereoProcessor::imageCb,
    Disparity + Pointcloud connected): pinned upload of both raws, the whole chain, stream sync.  Not part of `value`."""
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd import processor as P
    proc = P.GpuStereoProcessor(device)
    proc.initStereoModel(P.camera_info(W, H, **calib["left"]), P.camera_info(W, H, **calib["right"]))
    proc.setParams(params)
    proc.setMaxSpeckleSize(SPECKLE_SIZE)
    proc.setMaxSpeckleDiff(SPECKLE_DIFF)
    mask = capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD
    for _ in range(3):
        proc.processPair(lraw, rraw, ENCODING, mask)
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        proc.processPair(lraw, rraw, ENCODING, mask)
        ts.append(1e3 * (time.perf_counter() - t0))
    proc.close()
    ts.sort()
    return {"single_pair_ms_p50": round(ts[len(ts) // 2], 4), "single_pair_ms_p99": round(ts[min(len(ts) - 1, int(0.99 * len(ts)))], 4),
            "what": "gip_process_pair (imageCb schedule: upload L/R, mono, rectify L/R + colour L, StereoBM, speckle, reproject) for one %dx%d pair, host wall time incl. sync" % (W, H)}


E2E_MODES = None     # experiment hook (tools/e2e_probe.py): [(mode, frames_per_batch, compute_lanes), ...]
PCIE_GBS = 63.0  # /opt/skills/guides/MI355X_MICROARCH.md: host link PCIe Gen5 x16, 63 GB/s (spec, per direction)


def measured_link_gbs(device, mb=256, reps=6):
    """What the host link of this box moves: one large pinned copy per direction alone, and both directions at once (GB/s)."""
    import torch
    dev = torch.device("cuda", device)
    n = mb << 20
    hin, hout = torch.empty(n, dtype=torch.uint8).pin_memory(), torch.empty(n, dtype=torch.uint8).pin_memory()
    din, dout = torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def run(h2d, d2h):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            if h2d:
                with torch.cuda.stream(s1):
                    din.copy_(hin, non_blocking=True)
            if d2h:
                with torch.cuda.stream(s2):
                    hout.copy_(dout, non_blocking=True)
        torch.cuda.synchronize()
        return reps * n / (time.perf_counter() - t0) / 1e9
    run(True, True)
    a, b, c = run(True, False), run(False, True), run(True, True)
    return {"h2d": round(a, 1), "d2h": round(b, 1), "h2d_duplex": round(c, 1), "d2h_duplex": round(c, 1), "what": "%d MB pinned copies, %d per direction; duplex = each direction's rate with both running" % (mb, reps)}


def e2e_streaming(lraw, rraw, calib, params, device, budget_s=1.5):
    """End-to-end pairs/s through the streaming C-ABI (gip_streamer_*): every pair starts as two raw images in pageable host memory
    and ends as delivered outputs in pinned host memory -- pinned ring, copy-in stream, compute stream, copy-out stream, hipLaunchHostFunc
    delivery; the caller submits pair by pair.  Each record names its PCIe bound.  Not part of `value`."""
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd import processor as P
    from ros_gpu_stereo_processor_amd.streamer import Streamer
    px = W * H
    cn = 3 if ENCODING in ("bgr8", "rgb8") else 1
    proc = P.GpuStereoProcessor(device)
    proc.initStereoModel(P.camera_info(W, H, **calib["left"]), P.camera_info(W, H, **calib["right"]))
    proc.setParams(params)
    proc.setMaxSpeckleSize(SPECKLE_SIZE)
    proc.setMaxSpeckleDiff(SPECKLE_DIFF)
    out = []
    nsrc = len(lraw)
    for name, outputs, out_bytes in (("raw H2D + f32 disparity D2H", capi.TOPIC_DISPARITY, 4 * px),
                                     ("raw H2D + f32 disparity + 32-B PointCloud2 D2H", capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD, 36 * px)):
        # adaptive: the caller submits PAIR BY PAIR and the streamer forms the batches (a partly filled batch is launched whenever a compute lane is
        # free); fixed: full batches of 1 / 64 pairs only (the round-2 behaviour, for comparison)
        for mode, fpb, lanes in E2E_MODES if E2E_MODES else (("adaptive", 0, 1), ("fixed", 1, 1), ("fixed", 64, 1)):
            st = Streamer(proc, W, H, ENCODING, outputs, frames_per_batch=fpb, adaptive=(mode == "adaptive"), compute_lanes=lanes)
            warm = 200 if mode == "adaptive" else 4 * max(fpb, 1)
            for i in range(warm):          # warm-up: every ring slot used
                st.submit(lraw[i % nsrc], rraw[i % nsrc])
            st.flush()
            l0 = st.launches()
            n, t0 = 0, time.perf_counter()
            while True:
                for i in range(64):
                    st.submit(lraw[(n + i) % nsrc], rraw[(n + i) % nsrc])
                n += 64
                if time.perf_counter() - t0 > budget_s / 4:
                    break
            st.flush()
            dt = time.perf_counter() - t0
            assert st.delivered() == n + warm
            nl = st.launches() - l0
            st.close()
            in_bytes = 2 * px * cn
            bound = PCIE_GBS * 1e9 / max(in_bytes, out_bytes)     # full duplex: the larger direction binds
            out.append({"what": name, "batching": mode, "max_pairs_per_launch": fpb if fpb else "auto", "avg_pairs_per_launch": round(n / max(nl, 1), 2),
                        "host_input": "pageable, submitted pair by pair, copied into the pinned ring by the submitting thread",
                        "compute_lanes": lanes, "pairs": n, "pairs_per_s": round(n / dt, 1), "ms_per_pair": round(1e3 * dt / n, 4),
                        "h2d_bytes_per_pair": in_bytes, "d2h_bytes_per_pair": out_bytes, "pcie_bound_pairs_per_s": round(bound, 1),
                        "frac_of_pcie_bound": round(n / dt / bound, 4)})
    proc.close()
    link = measured_link_gbs(device)
    for r in out:
        # against what this box's link really moves in the binding direction (one large pinned copy per direction, both at once)
        r["frac_of_measured_link"] = round(r["pairs_per_s"] * max(r["h2d_bytes_per_pair"], r["d2h_bytes_per_pair"]) / 1e9 /
                                           (link["d2h_duplex"] if r["d2h_bytes_per_pair"] >= r["h2d_bytes_per_pair"] else link["h2d_duplex"]), 4)
    return {"measured_link_gbs": link, "api": "gip_streamer_open/submit/flush (pinned ring filled with streaming stores, copy-in / compute / copy-out streams, adaptive batching, hipLaunchHostFunc delivery)", "pcie_gbs_spec_per_direction": PCIE_GBS,
            "records": out}


def metric_name(wl):
    return "stereo pairs/sec + ms/frame disparity, %dx%d ndisp=%d at 1/2/4/8 GPU" % (wl["W"], wl["H"], wl["NDISP"])


def rowsplit_native(args, wl):
    """BASELINE configs[4] through the C-ABI (gip_rowsplit_*): ONE process owns one processor handle per tile (tile t on device t mod the
    visible devices), every buffer preallocated, halo rows by hipMemcpyPeerAsync, border components merged on the device.  A step is ONE
    frame.  For every tile count 1, 2, 4 ... up to --tiles: frames/s with two frames in flight (two tile sets taken in turn), the latency of
    one synchronised frame, and the time every tile needs with its device to itself (gip_rowsplit_run_serial: what a rank on its own GPU
    spends outside communication).  `value` is the frames/s of the largest tile count.  One frame is checked against the oracle chain."""
    import torch
    from ros_gpu_stereo_processor_amd import capi, rowsplit as RS
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    ndev = torch.cuda.device_count()
    tiles = max(1, args.tiles if args.tiles > 0 else args.gpus)
    counts = [n for n in (1, 2, 4, 8, 16) if n < tiles] + [tiles]
    extra_one_device = 4 if (tiles < 4 and not args.no_extra) else 0      # (c) of the round-3 brief: the 4-handle split on whatever devices there are
    lraw, rraw, calib = make_inputs(2, seed0=100, nseeds=2)
    params = capi.BMParams(**bm_kwargs())
    outputs = capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD
    dev_frames = {}

    def frames_on(d):
        if d not in dev_frames:
            dev_frames[d] = ([torch.from_numpy(x).to("cuda:%d" % d) for x in lraw], [torch.from_numpy(x).to("cuda:%d" % d) for x in rraw])
        return dev_frames[d]

    def measure(nt, depth=3, halo_mode=None, gather_mode=0):       # (three frames in flight: 0.54 ms per frame on one device, two: 0.60, four: 0.62)
        halo_mode = args.halo_mode if halo_mode is None else halo_mode
        devices = [t % ndev for t in range(nt)]
        groups = [RS.RowSplit(W, H, calib, params, nt, ENCODING, outputs, SPECKLE_SIZE, SPECKLE_DIFF, devices=devices, halo_mode=halo_mode, host_threads=args.host_threads,
                              gather_mode=gather_mode)
                  for _ in range(depth)]
        for k, g in enumerate(groups):          # group k always works on frame k mod 2: inputs resident in HBM before the timed region
            for t, d in enumerate(devices):
                L, R = frames_on(d)
                g.set_inputs(t, L[k % 2], R[k % 2])
        for i in range(args.warmup):
            groups[i % depth].run()
        for g in groups:
            g.wait()
        t0 = time.perf_counter()
        for i in range(args.steps):
            groups[i % depth].run()
        for g in groups:
            g.wait()
        dt = time.perf_counter() - t0
        lat = []
        for i in range(15):
            t1 = time.perf_counter()
            groups[0].run()
            groups[0].wait()
            lat.append(1e3 * (time.perf_counter() - t1))
        lat.sort()
        conc = [groups[0].last_timing(t) for t in range(nt)]
        # every tile's phases with the device to itself, median over several frames (the tile that goes first after a drain pays for the idle device;
        # run_serial rotates the order)
        runs = [groups[0].run_serial() for _ in range(9)]
        serial = [{k: sorted(r[t][k] for r in runs)[len(runs) // 2] for k in runs[0][t]} for t in range(nt)]
        noncomm = [round(x["front_ms"] + x["match_ms"] + x["post_ms"], 4) for x in serial]
        d16 = groups[0].download(want_f32=False, want_cloud=False)[0]
        for g in groups:
            g.close()
        rec = {"tiles": nt, "devices": devices, "halo": "every tile rectifies its own halo rows" if halo_mode else "hipMemcpyPeerAsync from the neighbours",
               "border_records": {0: "auto (gather kernel when the tiles share a device, else peer copies)", 1: "peer copies", 2: "gather kernel reading peer memory in place"}[gather_mode],
               "frames_in_flight": depth, "frames_per_s": round(args.steps / dt, 2), "ms_per_frame": round(1e3 * dt / args.steps, 4),
               "latency_ms_p50": round(lat[len(lat) // 2], 4), "per_tile_noncomm_ms": noncomm, "per_tile_noncomm_ms_max": max(noncomm),
               "per_tile_comm_ms_alone": [round(x["halo_ms"] + x["gather_ms"], 4) for x in serial],
               "per_tile_gpu_ms_concurrent": [round(x["total_ms"], 4) for x in conc],
               "per_tile_phases_alone_ms": [{k: round(v, 4) for k, v in x.items()} for x in serial]}
        return rec, d16

    records, verified, exp = [], None, None
    todo = [(nt, None, 0) for nt in counts + ([extra_one_device] if extra_one_device else [])]
    if args.variants and tiles > 1:
        # the other ways to move the halo rows and the border records, at the full tile count: which is faster is a property of the node's links
        todo += [(tiles, 1, 0), (tiles, 0, 2), (tiles, 1, 2)]
    for nt, hm, gm in todo:
        try:
            rec, d16 = measure(nt, halo_mode=hm, gather_mode=gm)
        except Exception as e:       # noqa: BLE001 -- a variant that does not run on this node must not take the other records down
            if hm is None and gm == 0:
                raise
            records.append({"tiles": nt, "halo_mode": hm, "gather_mode": gm, "error": repr(e)})
            continue
        if hm is not None or gm:
            rec["variant"] = True
        if nt == extra_one_device and nt not in counts:
            rec["note"] = "%d handles sharing %d device(s): concurrency without more hardware; the per-tile times alone are what a rank on its own GPU would see" % (nt, ndev)
        if not args.no_verify:
            if exp is None:
                from oracle import oracle as O
                maps = [O.init_undistort_rectify_map(calib[s_]["K"], calib[s_]["D"], calib[s_]["R"], calib[s_]["P"], W, H) for s_ in ("left", "right")]
                exp = cpu_chain(O, lraw[0], rraw[0], calib, maps, O.bm_params(**bm_kwargs()))[0]
            ok = bool(np.array_equal(d16, exp))
            rec["verified"] = ok
            verified = ok if verified is None else (verified and ok)
        records.append(rec)
    main_rec = [r for r in records if r["tiles"] == tiles and not r.get("variant") and "error" not in r][0]
    out = {"metric": metric_name(wl), "value": main_rec["frames_per_s"], "unit": "stereo pairs/s", "n_gpus": min(tiles, ndev), "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": main_rec["ms_per_frame"], "ms_per_frame": main_rec["ms_per_frame"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8",
           "data": "synthetic",
           "config": {"workload": wl["name"] + " -- ROW-TILE SPLIT of one frame over %d tile(s) in ONE process (gip_rowsplit_*): %s, border components merged on the device"
                                  % (tiles, "every tile rectifies its own halo rows" if args.halo_mode else "halo rows by hipMemcpyPeerAsync"),
                      "pairs_per_step": 1, "parallelism": "row-tile x%d on %d device(s), halo %d rows" % (tiles, min(tiles, ndev), RS.halo_rows(BLOCK))},
           "rowsplit": records, "verified": verified}
    print(json.dumps(out), flush=True)
    if verified is False:
        raise SystemExit("row split: the disparity of the timed configuration differs from the oracle")


def rowsplit_rccl(args, wl):
    """BASELINE configs[4], one process per GPU (strong scaling): every rank rectifies its own rows, swaps halo rows with its neighbours over
    RCCL point-to-point, matches its tile, takes part in the distributed speckle merge (one all-gather of border records, merged on the
    device) and produces float disparity + cloud for its rows.  `value` = frames per second of the whole job."""
    import torch
    import torch.distributed as dist
    from ros_gpu_stereo_processor_amd import capi, rowsplit as RS
    from ros_gpu_stereo_processor_amd import dist as gdist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # GIP_BENCH_REHEARSE=1: rehearsal of the N > 1 code path on a box with ONE GPU -- every rank uses device 0 and the collectives go
    # over gloo (RCCL refuses two ranks on one device).  Never set by the driver; the numbers of such a run mean nothing.
    rehearse = os.environ.get("GIP_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if rehearse:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    lraw, rraw, calib0 = make_inputs(2, seed0=100, nseeds=2)       # the same two frames on every rank (the raw frame is resident everywhere)
    calib, maps = gdist.broadcast_model(calib0 if rank == 0 else None, W, H, dev)
    params = capi.BMParams(**bm_kwargs())
    from ros_gpu_stereo_processor_amd.processor import GpuStereoProcessor, camera_info
    proc = GpuStereoProcessor(local_rank)
    proc.initStereoModel(camera_info(W, H, **calib["left"]), camera_info(W, H, **calib["right"]))
    Q = np.array(proc.model()["Q"], np.float64).reshape(4, 4)
    proc.close()
    L = [torch.from_numpy(x).to(dev) for x in lraw]
    R = [torch.from_numpy(x).to(dev) for x in rraw]

    def step(i):
        return RS.rowsplit_frame(L[i % 2], R[i % 2], maps, params, rank, world, SPECKLE_SIZE, int(SPECKLE_DIFF * 16), Q, calib["left"]["P"][2], calib["right"]["P"][2])
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    dist.barrier()
    dt = gdist.max_over_ranks(time.perf_counter() - t0, dev)
    if rank == 0:
        print(json.dumps({
            "metric": metric_name(wl), "value": round(args.steps / dt, 2), "unit": "stereo pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "ms_per_frame": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl["name"] + " -- ROW-TILE SPLIT of one frame over %d rank(s), one process per GPU: halo exchange (RCCL point-to-point) + all-gather of "
                                                "border records, merged on the device" % world,
                       "pairs_per_step": 1, "parallelism": "row-tile x%d, halo %d rows" % (world, RS.halo_rows(BLOCK))},
            "roofline": None, "note": "python-driven per-rank path (torch.distributed); the single-process form (bench.py --rowsplit without torchrun) is the preallocated one"}),
            flush=True)
    dist.destroy_process_group()


def rowsplit_main(args, wl):
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return rowsplit_rccl(args, wl)
    return rowsplit_native(args, wl)


def rowsplit_secondary(world, timeout_s=150):
    """The driver only ever runs `bench.py --gpus N`; so that a scaling run measures BASELINE configs[4] at all, rank 0 of an N >= 4 run
    starts the single-process row split (1, 2 and 4 tiles on devices 0..3) as a CHILD process once the frame-sharded measurement is over
    (the other ranks' devices are idle by then) and embeds its record.  A child with a time limit: whatever goes wrong there costs the
    secondary record only, never the main line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "uhd4k", "--rowsplit", "--tiles", "4", "--steps", "60", "--warmup", "6", "--no-extra", "--variants"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE",
                                                            "TORCHELASTIC_RUN_ID", "MASTER_PORT")}
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
        if lines:
            rec = json.loads(lines[-1])
            return {"workload": rec["config"]["workload"], "scaling": "strong", "unit": "frames/s", "records": rec.get("rowsplit"), "verified": rec.get("verified"),
                    "returncode": p.returncode}
        return {"error": "row-split child printed no record (rc %d): %s" % (p.returncode, (p.stderr or "")[-400:])}
    except subprocess.TimeoutExpired:
        return {"error": "row-split child exceeded %d s and was stopped" % timeout_s}
    except Exception as e:       # noqa: BLE001 -- the secondary record must never take the main line down
        return {"error": "row-split child failed: %r" % (e,)}


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--cpu-chain-only":   # child of cpu_baseline(): host cores only, no GPU, no torch
        n, dt, threads = _cpu_chain_timed(sys.argv[2], budget_s=float(sys.argv[3]) if len(sys.argv) > 3 else 6.0, max_pairs=100000)
        print(json.dumps({"pairs": n, "seconds": dt, "threads": threads}))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 200 steps (a quarter of a second of GPU time at the default workload): the timed region necessarily contains one pipeline fill and one
    # drain (about 1 ms with three batches in flight), which at 20 steps was 4 % of it
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="kitti", help="kitti = BASELINE configs[1] (the metric's config); hd1080 = configs[2]; uhd4k = the frame of configs[4] on one GPU")
    ap.add_argument("--batch", type=int, default=0, help="stereo pairs per step per GPU (default: 256 for kitti / refdefault, 64 for euroc, 8 for hd1080, 1 for uhd4k)")
    ap.add_argument("--depth", type=int, default=0, help="batches in flight per GPU (each on its own HIP stream with its own buffers): the pre/post "
                    "kernels of one batch run under the matcher of another.  Default per workload (kitti 3, others 2)")
    ap.add_argument("--matcher-stream", choices=["auto", "shared", "two", "own"], default="auto", help="shared: the matcher kernels of all batches in flight go to "
                    "ONE stream and run back to back (kitti, depth 3: 49.2k pairs/s against 43.2k with own streams); own: every batch keeps its "
                    "matcher on its own stream (better for the large-disparity workloads)")
    ap.add_argument("--schedule", choices=["auto", "batch", "fpm"], default="auto", help="batch: every batch in flight on a stream of its own (+ the shared matcher stream); "
                    "fpm: THREE streams in all -- every batch's front end in order on one, every batch's L/R check + speckle + outputs in order on another, "
                    "the matchers on the third (gip_batch_desc::phase), --depth handles in rotation")
    ap.add_argument("--fused-front-end", action="store_true", help="rectify + XSOBEL as ONE pass raw -> prefiltered plane (gipk_rectify_xsobel) instead of remap, then prefilter "
                    "(bit-identical; measured slower on MI355X, DESIGN.md 3.3)")
    ap.add_argument("--rowsplit", action="store_true", help="BASELINE configs[4] mode: ONE frame per step cut into row tiles over the ranks (halo exchange + distributed "
                    "speckle over RCCL); use with --workload uhd4k")
    ap.add_argument("--tiles", type=int, default=0, help="--rowsplit without torchrun: row tiles = processor handles in this one process (default: --gpus); tile t runs on device t mod "
                    "the visible devices")
    ap.add_argument("--halo-mode", type=int, default=0, choices=[0, 1], help="--rowsplit: 0 = neighbours exchange rectified halo rows (hipMemcpyPeerAsync), 1 = every tile rectifies its own halo rows")
    ap.add_argument("--host-threads", type=int, default=0, choices=[0, 1], help="--rowsplit: 0 = one enqueueing host thread per tile, 1 = the calling thread enqueues everything")
    ap.add_argument("--variants", action="store_true", help="--rowsplit: also measure the other halo / border-record transports at the full tile count")
    ap.add_argument("--no-extra", action="store_true", help="--rowsplit: skip the additional 4-handle record on fewer than 4 devices")
    ap.add_argument("--no-verify", action="store_true", help="skip the comparison of one frame of the timed configuration with the oracle chain")
    ap.add_argument("--no-secondary", action="store_true", help="only the metric's own workload: no hd1080 / refdefault / utest_timing records, no row-split child at N >= 4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--plain-streams", action="store_true", help="EXPERIMENT: create the pipeline's streams with torch.cuda.Stream in program order instead of gip_concurrent_streams_create")
    ap.add_argument("--chain-priority", type=int, default=0, help="HIP stream priority of the per-batch streams that carry everything but the matcher (-1 = high)")
    ap.add_argument("--matcher-priority", type=int, default=0, help="HIP stream priority of the shared matcher stream (negative = higher).  Measured: -1 starves the kernels that feed the matcher, 41.8 k against 53.4 k pairs/s at 0")
    ap.add_argument("--ablate", default="", help="experiment (NOT the metric's configuration; implies --no-verify): comma list of stages to leave out -- speckle, f32 (with --no-cloud: no output kernel at all), lr (no L/R check)")
    ap.add_argument("--no-cloud", action="store_true", help="experiment (NOT the metric's configuration): float disparity only, no PointCloud2 records")
    ap.add_argument("--chain-cu-mask", default="", help="experiment: PATTERN/PERIOD, e.g. 0x1/4 = the per-batch (non-matcher) streams may use every fourth CU only")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-pair latency probe (keeps kernel profiles clean)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end streaming measurement (host memory in, host memory out)")
    ap.add_argument("--ndisp", type=int, default=0, help="experiment: override the workload's numDisparities (the metric's configuration is the default)")
    ap.add_argument("--block", type=int, default=0, help="experiment: override the workload's block size")
    args = ap.parse_args()
    global W, H, NDISP, BLOCK, ENCODING
    wl = dict(WORKLOADS[args.workload])
    if args.ndisp > 0 or args.block > 0:
        wl["NDISP"] = args.ndisp if args.ndisp > 0 else wl["NDISP"]
        wl["BLOCK"] = args.block if args.block > 0 else wl["BLOCK"]
        wl["name"] = wl["name"] + " [EXPERIMENT: ndisp=%d block=%d]" % (wl["NDISP"], wl["BLOCK"])
        wl["kernel"] = "k_bm_c4"
    W, H, NDISP, BLOCK, ENCODING = wl["W"], wl["H"], wl["NDISP"], wl["BLOCK"], wl["ENCODING"]
    if args.rowsplit:
        return rowsplit_main(args, wl)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # GIP_BENCH_REHEARSE=1: rehearsal of the N > 1 code path on a box with ONE GPU -- every rank uses device 0 and the collectives go
    # over gloo (RCCL refuses two ranks on one device).  Never set by the driver; the numbers of such a run mean nothing.
    rehearse = os.environ.get("GIP_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ctx = dict(torch=torch, dist=dist, world=world, rank=rank, local_rank=local_rank, dev=dev)
    out = chain_workload(args, args.workload, wl, ctx, args.steps, args.warmup, args.batch, args.depth, probes=True)
    if rank == 0:
        # the other headline shapes on the same record (N = 1 only): north_star's second size (BASELINE configs[2]), the reference constructor's
        # own matcher shape, and the shape of the reference's only timing harness -- each with value, roofline and CPU baseline
        if world == 1 and args.workload == "kitti" and not args.no_secondary and args.ndisp == 0 and args.block == 0:
            sec = []
            for name, steps in (("hd1080", 60), ("refdefault", 100)):
                wl2 = dict(WORKLOADS[name])
                set_workload(wl2)
                sec.append(chain_workload(args, name, wl2, ctx, steps, 6, 0, 0, probes=False))
            sec.append(utest_timing_record(args, ctx))
            set_workload(wl)
            out["secondary"] = sec
            out["verified"] = bool(out.get("verified")) and all(r.get("verified") is not False for r in sec)
        # ... and BASELINE configs[4] (row-tile split of one 4096x2160 frame) when there are devices to split over
        # (GIP_BENCH_FORCE_ROWSPLIT=1: rehearsal of this branch on a box with fewer devices -- the tiles then share devices)
        if ((world >= 4 and not rehearse) or os.environ.get("GIP_BENCH_FORCE_ROWSPLIT") == "1") and not args.no_secondary and os.environ.get("GIP_BENCH_NO_ROWSPLIT") != "1":
            out["rowsplit"] = None       # filled in below, once the other ranks have left their devices
    # the frame-sharded job ends here for everybody; the other ranks release their GPUs before rank 0 starts the row-split child
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if "rowsplit" in out:
            time.sleep(2.0)          # let the other ranks exit
            out["rowsplit"] = rowsplit_secondary(world)
        print(json.dumps(out), flush=True)
        if out.get("verified") is False:
            raise SystemExit("bench.py: an output of the timed configuration differs from the oracle chain")


def set_workload(wl):
    global W, H, NDISP, BLOCK, ENCODING
    W, H, NDISP, BLOCK, ENCODING = wl["W"], wl["H"], wl["NDISP"], wl["BLOCK"], wl["ENCODING"]


def chain_workload(args, key, wl, ctx, steps, warmup, batch, depth, probes):
    """One chain workload (the globals W, H, ... are set to it): returns the record on rank 0 (None elsewhere).  probes: the main line's
    extras (single-pair latency, end-to-end streaming); the CPU baseline and the oracle check of the timed configuration are always there."""
    torch, dist, world, rank, local_rank, dev = ctx["torch"], ctx["dist"], ctx["world"], ctx["rank"], ctx["local_rank"], ctx["dev"]
    from ros_gpu_stereo_processor_amd import capi
    from ros_gpu_stereo_processor_amd import dist as gdist
    from ros_gpu_stereo_processor_amd.pipeline import BatchPipeline
    B = batch if batch > 0 else wl["batch"]
    depth = depth if depth > 0 else wl["depth"]
    shared_matcher = wl["shared"] if args.matcher_stream == "auto" else args.matcher_stream == "shared"
    schedule = wl.get("schedule", "batch") if args.schedule == "auto" else args.schedule
    if schedule == "fpm":
        shared_matcher = True
    # BASELINE configs[3]: a synthetic bag of BAG_FRAMES stereo pairs (30 s at 30 fps), frame g -> rank g mod N (dist.frame_shard), no
    # data-path collective.  A step is one batch of B pairs PER RANK, so step i of rank r holds the bag frames
    # ((i * B + j) * N + r) mod BAG_FRAMES, j = 0..B-1: N ranks together walk the bag front to back, each on its own shard
    # ("weak" scaling: the per-GPU work of a step does not depend on N).  The bag cycles through NSCENES distinct synthetic scenes.
    BAG_FRAMES, NSCENES = 900, (8 if probes else 4)
    lscene, rscene, calib0 = make_inputs(NSCENES, seed0=100, nseeds=NSCENES)
    lraw, rraw = lscene, rscene     # host copies (latency / e2e probes, oracle check)
    # Calibration + rectification tables: rank 0 owns the CameraInfo, builds the fixed-point tables on its host (double
    # precision) and broadcasts both once over RCCL (xGMI); the other ranks run on exactly those tables.  Frames are
    # independent units (frame i -> rank i mod N, dist.frame_shard), so no collective follows on the data path.
    calib, maps = gdist.broadcast_model(calib0 if rank == 0 else None, W, H, dev)
    ablate = set(x for x in args.ablate.split(",") if x)
    if ablate:
        args.no_verify = True
    if "lr" in ablate:
        global DISP12
        DISP12 = -1
    params = capi.BMParams(**bm_kwargs())
    outputs = capi.TOPIC_DISPARITY | (0 if args.no_cloud else capi.TOPIC_POINTCLOUD)
    pipes = [BatchPipeline(W, H, calib, params, B, ENCODING, outputs, local_rank, 0 if "speckle" in ablate else SPECKLE_SIZE, SPECKLE_DIFF, want_f32="f32" not in ablate, fused_front_end=args.fused_front_end, batches_in_flight=max(1, depth))
             for _ in range(max(1, depth))]
    if args.chain_cu_mask:
        # experiment: the per-batch streams (everything but the matcher) restricted to a subset of the CUs (hipExtStreamCreateWithCUMask), so that the
        # matcher keeps the other CUs to itself.  The mask is a bit string repeated over the 256 CUs, e.g. 0x1 period 4 = every fourth CU.
        hip = C.CDLL("libamdhip64.so")
        pattern, period = [int(x, 0) for x in args.chain_cu_mask.split("/")]
        bits = 0
        for i in range(256):
            if (pattern >> (i % period)) & 1:
                bits |= 1 << i
        words = (C.c_uint32 * 8)(*[(bits >> (32 * k)) & 0xFFFFFFFF for k in range(8)])
        streams = []
        for _ in pipes:
            hs = C.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(C.byref(hs), 8, words)
            assert rc == 0, "hipExtStreamCreateWithCUMask failed: %d" % rc
            streams.append(torch.cuda.ExternalStream(hs.value, device=dev))
    elif args.plain_streams or args.chain_priority or args.matcher_priority:
        streams = [torch.cuda.Stream(device=dev, priority=args.chain_priority) for _ in pipes]
    else:
        streams = None
    want_matcher = len(pipes) > 1 and shared_matcher
    independent_streams = None
    if streams is None:
        # the batch streams and the matcher stream as ONE set of streams that were measured to dispatch side by side (gip_concurrent_streams_create):
        # which hardware queue a HIP stream lands on depends on every stream the process created before, and a matcher stream that shares a
        # queue with a batch stream costs 12-25 %
        two = args.matcher_stream == "two" and len(pipes) > 1
        nbatch_streams = 2 if (schedule == "fpm" and want_matcher) else len(pipes)      # fpm: one stream for the front ends, one for the back ends
        handles, independent_streams = capi.concurrent_streams(nbatch_streams + (2 if two else 1 if want_matcher else 0), local_rank)
        ext = [torch.cuda.ExternalStream(hd, device=dev) for hd in handles]
        streams, matcher_stream = ext[:nbatch_streams], (ext[nbatch_streams] if (want_matcher or two) else None)
    else:
        matcher_stream = torch.cuda.Stream(device=dev, priority=args.matcher_priority) if want_matcher else None
    # "two": two matcher streams taken in turn, so that the ragged end of one matcher launch overlaps the start of the next
    matcher_streams = [matcher_stream]
    if args.matcher_stream == "two" and len(pipes) > 1:
        if independent_streams is not None:
            matcher_streams = [ext[len(pipes)], ext[len(pipes) + 1]]
        else:
            matcher_stream = torch.cuda.Stream(device=dev)
            matcher_streams = [matcher_stream, torch.cuda.Stream(device=dev)]
    for p in pipes:
        p.install_maps(maps)

    # every step's input batch is resident in HBM before the timed region starts (gathered from the scenes on the device)
    scenes_l, scenes_r = torch.from_numpy(lscene).to(dev), torch.from_numpy(rscene).to(dev)
    j = torch.arange(B, device=dev)

    def scene_of(i, jj):
        return (((i * B + jj) * world + rank) % BAG_FRAMES) % NSCENES

    def step_inputs(i):
        idx = (((i * B + j) * world + rank) % BAG_FRAMES) % NSCENES
        return scenes_l.index_select(0, idx).contiguous(), scenes_r.index_select(0, idx).contiguous()
    nsteps_total = warmup + steps
    ninputs = min(nsteps_total, 32)                                # 32 distinct batches (3.8 GB at the default shape), then they repeat
    inputs = [step_inputs(i) for i in range(ninputs)]
    left, right = inputs[0]

    fpm = schedule == "fpm" and len(pipes) > 1 and matcher_stream is not None

    def run_step(i):
        k = i % len(pipes)
        l, r = inputs[i % len(inputs)]
        if fpm:
            with torch.cuda.stream(streams[0]):
                pipes[k].run(l, r, matcher_stream, phase=capi.BATCH_FRONT)
            with torch.cuda.stream(streams[1]):
                pipes[k].run(l, r, matcher_stream, phase=capi.BATCH_BACK)
            return
        with torch.cuda.stream(streams[k]):
            pipes[k].run(l, r, matcher_streams[i % len(matcher_streams)])
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    bm_alone_ms = None
    if len(pipes) > 1:
        # the dominant kernel on its own (no second batch in flight): three synchronised steps of pipeline 0, declared as such
        pipes[0].batches_in_flight = 1
        pipes[0].run(left, right)
        torch.cuda.synchronize()
        pipes[0].timing_reset()
        for _ in range(3):
            pipes[0].run(left, right)
            torch.cuda.synchronize()
        a_bm, _, a_n = pipes[0].timing_read()
        bm_alone_ms = a_bm / max(a_n, 1)
        pipes[0].batches_in_flight = len(pipes)
    for i in range(warmup):
        run_step(i)
    torch.cuda.synchronize()
    for p in pipes:
        p.timing_reset()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        run_step(warmup + i)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    bm_ms_sum, total_ms_sum, span_ms_sum, runs = [sum(x) for x in zip(*[p.timing_read_spans() for p in pipes])]

    dt_max = gdist.max_over_ranks(dt, dev)

    out = None
    if rank == 0:
        pairs = world * B * steps
        value = pairs / dt_max
        bm_ms = bm_ms_sum / max(runs, 1)
        if len(pipes) == 1:
            bm_alone_ms = bm_ms
        # dominant kernel: k_bm_c4 (SAD block matcher on the prefiltered rectified pair).  Algorithmic bytes per
        # stereo pair = 4*W*H (prefiltered L + R read once, int16 disparity written once; SURVEY.md section 8d,
        # "SAD-only kernel on rectified input"); one launch processes B pairs.
        algo_bytes = 4.0 * W * H * B
        achieved = algo_bytes / (bm_ms * 1e-3) / 1e9 if bm_ms > 0 else 0.0
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath)).get("workloads", {}).get(key)
                if tj and tj.get("batch") == B:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = "static: profiles/traffic.json (2 x FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc passes of this workload, not measured in this run)"
            except Exception:
                traffic = None
        # the timed configuration checked against the oracle (outside the timing): frame 0 of the LAST step every pipeline ran, the whole chain
        verified = None
        if not args.no_verify:
            from oracle import oracle as O
            omaps = [O.init_undistort_rectify_map(calib[s_]["K"], calib[s_]["D"], calib[s_]["R"], calib[s_]["P"], W, H) for s_ in ("left", "right")]
            po = O.bm_params(**bm_kwargs())
            verified, cache = True, {}
            for k in range(len(pipes)):
                last = [i for i in range(warmup + steps) if i % len(pipes) == k]
                if not last:
                    continue
                sc = scene_of(last[-1] % ninputs, 0)
                if sc not in cache:
                    cache[sc] = cpu_chain(O, lraw[sc], rraw[sc], calib, omaps, po)
                e16, epc = cache[sc]
                g16 = pipes[k].disparity[0].cpu().numpy()
                if pipes[k].pointcloud is None:
                    verified = verified and bool(np.array_equal(g16, e16))
                    continue
                gpc = pipes[k].pointcloud[0].cpu().numpy()
                gx, ex = gpc[..., :12].copy().view(np.float32), epc[..., :12].copy().view(np.float32)
                ok = np.array_equal(g16, e16) and np.array_equal(np.isnan(gx), np.isnan(ex)) and np.array_equal(gpc[..., 12:], epc[..., 12:])
                fin = ~np.isnan(ex)
                ok = ok and bool(np.allclose(gx[fin], ex[fin], rtol=2.4e-7, atol=0))      # float xyz: <= 1 ulp, as in tests/test_gpu_pipeline.py
                ok = ok and np.array_equal(pipes[k].disparity_f32[0].cpu().numpy(), O.disparity_to_f32(e16, calib["left"]["P"][2], calib["right"]["P"][2]))
                verified = verified and bool(ok)
        out = {
            "metric": metric_name(wl),
            "value": round(value, 2), "unit": "stereo pairs/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(1e3 * dt_max / steps, 4), "ms_per_frame": round(1e3 * dt_max / (B * steps), 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": wl["name"] + (" [EXPERIMENT: no PointCloud2 output]" if args.no_cloud else "") + (" [EXPERIMENT: without %s]" % args.ablate if args.ablate else ""), "pairs_per_step_per_gpu": B, "batches_in_flight_per_gpu": len(pipes), "schedule": ("front / back / matcher streams" if fpm else "one stream per batch"), "independent_streams": independent_streams, "matcher_stream": ("two" if len(matcher_streams) > 1 else "shared") if matcher_stream is not None else "own",
                       "parallelism": "frame-sharded x%d: frame g of a %d-frame bag -> rank g mod %d" % (world, BAG_FRAMES, world)},
            "verified": verified,
            "roofline": {"bound": "hbm", "kernel": wl["kernel"], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source, "kernel_ms": round(bm_ms, 4),
                         "algorithmic_bytes_per_launch": algo_bytes, "chain_gpu_ms_per_step": round(total_ms_sum / max(runs, 1), 4),
                         # kernel_ms is the launch duration inside the timed region, where the kernels of the other batch in flight
                         # share the GPU; kernel_ms_alone is the same launch with nothing else running (three untimed steps)
                         # (launched on the caller's own stream; for one-wave shapes with block >= 11 that is the matcher's refetch form,
                         # inside the pipeline its ring form -- two stagings of the same kernel; identical at the metric's shape)
                         "kernel_ms_alone": round(bm_alone_ms, 4) if bm_alone_ms else None,
                         # the search is integer-VALU / latency bound, not HBM bound: (pixel, disparity) pairs per
                         # second times a FIXED algorithmic floor of issue slots per pair, against the measured VALU issue rate
                         "valu_algorithmic": ({"px_disp_pairs_per_s": round(W * H * NDISP * B / (bm_alone_ms * 1e-3), 1), "floor_slots_per_px_disp": VALU_FLOOR_SLOTS,
                                               "peak_lane_slots_per_s": VALU_PEAK_LANE_SLOTS, "from": "kernel_ms_alone",
                                               "frac": round(W * H * NDISP * B / (bm_alone_ms * 1e-3) * VALU_FLOOR_SLOTS / VALU_PEAK_LANE_SLOTS, 4)} if bm_alone_ms else None),
                         # SURVEY.md 8(d)'s figure for "the remap+SAD kernel": 16 bytes per pixel (2 raw + 12 map + 2 disparity), here over the SPAN
                         # of the launches that do that work (rectify x2, prefilter x2, matcher, validateDisparity), events on the batch's stream
                         "remap_sad_span": {"algorithmic_bytes_per_launch_set": 16.0 * W * H * B, "span_ms": round(span_ms_sum / max(runs, 1), 4),
                                            "achieved": round(16.0 * W * H * B / (span_ms_sum / max(runs, 1) * 1e-3) / 1e9, 2) if span_ms_sum > 0 else None,
                                            "frac": round(16.0 * W * H * B / (span_ms_sum / max(runs, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if span_ms_sum > 0 else None,
                                            "note": "not one kernel: remap inside the matcher would re-gather every pixel 3-4x in the VALU-critical kernel (DESIGN.md 3.3)"}},
        }
        if world == 1 and probes and not args.no_latency:
            out["latency"] = single_pair_latency(lraw[0], rraw[0], calib, params, local_rank)
        if world == 1 and probes and not args.no_e2e:
            out["e2e"] = e2e_streaming(lraw, rraw, calib, params, local_rank)
    for p in pipes:
        p.close()
    del pipes, inputs, scenes_l, scenes_r
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(key, budget_s=5.0 if probes else 3.0)
    return out


def utest_timing_record(args, ctx, B=64, iters=60):
    """The shape of the reference's only timing harness, DisparityTiming (reference test/UTest.cpp:400-429): 752x480, createStereoBM(48, 19),
    the MATCHER ONLY on an already rectified pair (computeDisparityBare = block_matcher->compute: prefilter + match + L/R check + blanking),
    here on B resident pairs per launch set.  CPU baseline = the oracle's cv::StereoBM::compute restatement on the same pairs."""
    torch, dev = ctx["torch"], ctx["dev"]
    from ros_gpu_stereo_processor_amd import kernels as K, synth
    from oracle import oracle as O
    w, h, nd, block = 752, 480, 48, 19
    pairs = [synth.rectified_pair(w, h, nd, seed=300 + i)[:2] for i in range(4)]
    L = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(B)])).to(dev)
    R = torch.from_numpy(np.stack([pairs[i % 4][1] for i in range(B)])).to(dev)
    kw = dict(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=block, min_disparity=0, num_disparities=nd, texture_threshold=10, uniqueness_ratio=15,
              disp12_max_diff=1)
    p = K.bm_params(**kw)
    for _ in range(5):
        d = K.stereobm(L, R, p)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(iters):
        d = K.stereobm(L, R, p)
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpu_ms = e0.elapsed_time(e1) / iters
    po = O.bm_params(**kw)
    verified = None
    if not args.no_verify:
        got = d.cpu().numpy()
        verified = all(bool(np.array_equal(got[i], O.stereobm(pairs[i][0], pairs[i][1], po, nthreads=1))) for i in range(4))
    rec = {"workload": "utest_timing: 752x480 mono8 rectified pairs, ndisp=48 block=19, MATCHER ONLY (cv::StereoBM::compute: XSOBEL prefilter + SAD/WTA/uniqueness/sub-pixel + L/R check "
                       "+ ROI blanking) -- the shape of the reference's DisparityTiming test (test/UTest.cpp:400-429)",
           "value": round(B * iters / dt, 2), "unit": "stereo pairs/s", "ms_per_step": round(1e3 * dt / iters, 4), "pairs_per_step": B, "steps": iters, "verified": verified,
           "roofline": {"bound": "hbm", "kernel": "gipk_stereobm launch set (k_prefilter_xsobel_pk x2, k_bm_c4<3,9,1>, k_bm_finalize_r)", "algorithmic_bytes_per_launch": 4.0 * w * h * B,
                        "kernel_ms": round(gpu_ms, 4), "achieved": round(4.0 * w * h * B / (gpu_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(4.0 * w * h * B / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                        "valu_algorithmic": {"px_disp_pairs_per_s": round(w * h * nd * B / (gpu_ms * 1e-3), 1), "floor_slots_per_px_disp": VALU_FLOOR_SLOTS,
                                             "peak_lane_slots_per_s": VALU_PEAK_LANE_SLOTS, "from": "kernel_ms (the whole launch set)",
                                             "frac": round(w * h * nd * B / (gpu_ms * 1e-3) * VALU_FLOOR_SLOTS / VALU_PEAK_LANE_SLOTS, 4)}}}
    if not args.no_cpu_baseline:
        n, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 3.0:
            O.stereobm(pairs[n % 4][0], pairs[n % 4][1], po, nthreads=1)
            n += 1
        secs = time.perf_counter() - t1
        rec["cpu_baseline"] = {"value": round(n / secs, 2), "unit": "stereo pairs/s", "cores": 1, "kind": "port",
                               "sample": "%d calls of the oracle's cv::StereoBM::compute restatement on the same 752x480 pairs in %.1f s, one thread" % (n, secs), "cpu_model": _cpu_model()}
    return rec


if __name__ == "__main__":
    main()
