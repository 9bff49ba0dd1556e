#!/usr/bin/env python3
"""Two (or more) ranks over RCCL (or, with GIP_BENCH_REHEARSE=1, over gloo with every rank on device 0): (1) dist.broadcast_model, every rank must end with rank 0's tables; (2) the row-tile split of one frame
(rowsplit.rowsplit_frame: halo exchange by batched isend/irecv, border all-gather for the speckle merge) must reproduce the single-GPU
disparity.  Launched by tests/test_gpu_pipeline.py::test_nccl_broadcast_and_halo_exchange_two_devices through torch.distributed.run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ros_gpu_stereo_processor_amd import capi, synth, kernels as K, rowsplit as RS  # noqa: E402
from ros_gpu_stereo_processor_amd import dist as D  # noqa: E402


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    rehearse = os.environ.get("GIP_BENCH_REHEARSE") == "1"     # one-GPU box: every rank on device 0, collectives over gloo
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if rehearse:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    w, h, nd = 640, 240, 64
    calib0 = synth.scaled_calibration(w, h)
    calib, maps = D.broadcast_model(calib0 if rank == 0 else None, w, h, dev)
    for i, side in enumerate(("left", "right")):
        m1, m2 = D.build_maps_host(calib0[side], w, h)
        assert np.array_equal(maps[2 * i].cpu().numpy(), m1) and np.array_equal(maps[2 * i + 1].cpu().numpy(), m2.view(np.int16)), "broadcast tables differ"
    lr, rr, _, _ = synth.raw_pair(w, h, nd, seed=21, calib=calib0)
    kw = dict(prefilter_type=1, prefilter_size=9, prefilter_cap=31, block_size=15, min_disparity=0, num_disparities=nd, texture_threshold=10, uniqueness_ratio=15,
              disp12_max_diff=1, speckle_window_size=0, speckle_range=0)
    p = capi.BMParams(**kw)
    tl, tr = torch.from_numpy(lr).to(dev), torch.from_numpy(rr).to(dev)
    y0, y1, d, _, _ = RS.rowsplit_frame(tl, tr, maps, p, rank, world, 200, 32, want_cloud=False)
    full = K.filter_speckles(K.stereobm(K.remap(tl, maps[0], maps[1]), K.remap(tr, maps[2], maps[3]), p).clone(), -16, 200, 32)
    assert torch.equal(d, full[y0:y1]), "row-tile result differs from the single-GPU frame on rank %d" % rank
    dist.barrier()
    if rank == 0:
        print("%s selftest ok: %d ranks, %dx%d" % ("gloo-on-one-GPU" if rehearse else "nccl", world, w, h))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
