#!/usr/bin/env python3
"""BASELINE.json configs[4]: ONE 4096x2160 mono8 pair, ndisp 192, block 15, split into row tiles across the ranks with a
halo exchange over RCCL (xGMI), result gathered on rank 0 (speckle filter + float disparity there).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 tools/rowsplit_demo.py [--check]

Every rank uploads the raw pair (8.8 MB per eye), rectifies only the rows it owns plus nothing else (the remap is a gather, so
it needs the whole raw frame but writes only its tile), swaps halo rows of the RECTIFIED images with its neighbours, matches its
extended tile and sends its rows to rank 0.  Prints ms/frame (max over ranks) and, with --check, compares with the 1-GPU result.
"""
import argparse, os, sys, time
import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ros_gpu_stereo_processor_amd import kernels as K, rowsplit as RS, synth, dist as gdist  # noqa: E402

W, H, NDISP, BLOCK = 4096, 2160, 192, 15


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    calib0 = synth.scaled_calibration(W, H) if rank == 0 else None
    calib, maps = gdist.broadcast_model(calib0, W, H, dev)
    lraw, rraw, _, _ = synth.raw_pair(W, H, NDISP, seed=500, calib=calib)   # every rank generates the same frame (stands in for the upload)
    lraw, rraw = torch.from_numpy(lraw).to(dev), torch.from_numpy(rraw).to(dev)
    p = K.bm_params(block_size=BLOCK, num_disparities=NDISP, texture_threshold=10, uniqueness_ratio=15, disp12_max_diff=1)
    ranges = RS.plan(H, world)
    halo = RS.halo_rows(BLOCK)
    y0, y1 = ranges[rank]

    def frame():
        # rectify only this rank's rows: the tables are sliced, the source is the whole raw frame
        rl = K.remap(lraw, maps[0][y0:y1].contiguous(), maps[1][y0:y1].contiguous())
        rr = K.remap(rraw, maps[2][y0:y1].contiguous(), maps[3][y0:y1].contiguous())
        if world > 1:
            rl = RS.exchange_halos(rl, rank, world, ranges, H, halo)
            rr = RS.exchange_halos(rr, rank, world, ranges, H, halo)
        d = RS.crop_valid(RS.match_tile(rl, rr, p), H, y0, y1, halo) if world > 1 else RS.match_tile(rl, rr, p)
        full = RS.gather_rows(d, rank, world, ranges) if world > 1 else d
        if rank == 0:
            K.filter_speckles(full, -16, 800, 80)
        return full

    out = frame()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        out = frame()
    torch.cuda.synchronize()
    dt = gdist.max_over_ranks((time.perf_counter() - t0) / args.iters, dev)
    if rank == 0:
        print("rowsplit %dx%d ndisp %d block %d on %d GPU(s): %.3f ms/frame" % (W, H, NDISP, BLOCK, world, dt * 1e3))
        if args.check:
            rl = K.remap(lraw, maps[0], maps[1]); rr = K.remap(rraw, maps[2], maps[3])
            ref = K.filter_speckles(K.stereobm(rl, rr, p), -16, 800, 80)
            print("identical to the single-GPU result:", bool(torch.equal(ref, out)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
