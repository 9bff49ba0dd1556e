"""CPU test of the N>1 path: world_size 2 over gloo -- broadcast of CameraInfo + rectification tables from rank 0, frame
sharding (independent units, no data-path collective), max-over-ranks timing."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ros_gpu_stereo_processor_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ros_gpu_stereo_processor_amd import dist as D
    w, h = 320, 200
    calib = synth.scaled_calibration(w, h) if rank == 0 else None   # only rank 0 knows the calibration
    got_calib, maps = D.broadcast_model(calib, w, h, torch.device("cpu"))
    # every rank can rebuild the tables from the broadcast CameraInfo and must get what rank 0 sent
    same = True
    for i, side in enumerate(("left", "right")):
        m1, m2 = D.build_maps_host(got_calib[side], w, h)
        same &= np.array_equal(m1, maps[2 * i].numpy()) and np.array_equal(m2.view(np.int16), maps[2 * i + 1].numpy())
    shard = D.frame_shard(rank, world, 11)
    tmax = D.max_over_ranks(1.0 + rank, torch.device("cpu"))
    q.put((rank, same, shard, tmax, float(got_calib["left"]["K"][0]), maps[1].numpy().astype(np.int64).sum()))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2(oracle):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)                       # tables identical on every rank
    assert sorted(res[0][2] + res[1][2]) == list(range(11)) and not set(res[0][2]) & set(res[1][2])
    assert res[0][3] == res[1][3] == 2.0                # max over ranks
    assert res[0][4] == res[1][4] and res[0][5] == res[1][5]
    # and they are the oracle's tables
    calib = synth.scaled_calibration(320, 200)
    from ros_gpu_stereo_processor_amd import dist as D
    m1, m2 = D.build_maps_host(calib["left"], 320, 200)
    e1, e2 = oracle.init_undistort_rectify_map(calib["left"]["K"], calib["left"]["D"], calib["left"]["R"], calib["left"]["P"], 320, 200)
    assert np.array_equal(m1, e1) and np.array_equal(m2, e2)


def _rowsplit_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ros_gpu_stereo_processor_amd import rowsplit as RS
    h, w, halo = 37, 16, 5
    full = torch.arange(h * w, dtype=torch.int32).reshape(h, w).to(torch.uint8)       # every rank can rebuild the truth
    ranges = RS.plan(h, world)
    y0, y1 = ranges[rank]
    ext = RS.exchange_halos(full[y0:y1].clone(), rank, world, ranges, h, halo)
    e0, e1 = RS.extended_range(h, y0, y1, halo)
    ok = bool(torch.equal(ext, full[e0:e1])) and e0 % 2 == 0
    # "compute" = identity on the extended tile, crop, gather on rank 0
    mine = RS.crop_valid(ext.to(torch.int16), h, y0, y1, halo)
    stitched = RS.gather_rows(mine, rank, world, ranges)
    if rank == 0:
        ok = ok and bool(torch.equal(stitched, full.to(torch.int16)))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_rowsplit_halo_exchange_world3():
    """config 5 plumbing on CPU: halo exchange between neighbouring row tiles and gather of the result tiles (gloo, 3 ranks)."""
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rowsplit_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)
