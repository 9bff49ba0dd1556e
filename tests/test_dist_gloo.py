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


class _CpuLabels:
    """Host stand-in for kernels.SpeckleLabels so that the EXCHANGE of rowsplit.distributed_speckle (one all-gather of border records in
    the layout of k_spk_border_record) runs over gloo without a GPU: components of ONE tile by flood fill with cv::filterSpeckles'
    connectivity rule (roots are arbitrary tile-local ids), border record and border merge restated in numpy.  Test infrastructure only:
    the product labeler does both steps on the device (tests/test_gpu_rowsplit.py, test_gpu_pipeline.py check it against the oracle)."""

    def __init__(self, tile, new_val, max_diff):
        self.tile, self.new_val = tile, new_val
        a = tile.numpy().astype(np.int64)
        h, w = a.shape
        self.lab = -np.ones((h, w), np.int64)
        self.sizes = {}
        nxt = 0
        for y in range(h):
            for x in range(w):
                if a[y, x] == new_val or self.lab[y, x] >= 0:
                    continue
                rid = 1000 + 7 * nxt      # deliberately not dense / not zero based
                nxt += 1
                stack, n = [(y, x)], 0
                self.lab[y, x] = rid
                while stack:
                    cy, cx = stack.pop()
                    n += 1
                    for ny, nx in ((cy - 1, cx), (cy + 1, cx), (cy, cx - 1), (cy, cx + 1)):
                        if 0 <= ny < h and 0 <= nx < w and self.lab[ny, nx] < 0 and a[ny, nx] != new_val and abs(a[ny, nx] - a[cy, cx]) <= max_diff:
                            self.lab[ny, nx] = rid
                            stack.append((ny, nx))
                self.sizes[rid] = n

    def _rows(self):
        h = self.tile.shape[0]
        return (0, h - 1)

    def border_record(self, tile):
        """the record layout of k_spk_border_record (rowsplit.border_record_layout), filled on the host"""
        from ros_gpu_stereo_processor_amd import rowsplit as RS
        w = self.tile.shape[1]
        lay = RS.border_record_layout(w)
        S = lay["nbytes"] // 4
        lab = np.concatenate([self.lab[r] for r in self._rows()])
        val = np.concatenate([self.tile.numpy()[r] for r in self._rows()]).astype(np.int16)
        par = (tile * S + w + np.arange(2 * w)).astype(np.int32)       # invalid pixels point at themselves
        size = np.zeros(2 * w, np.int32)
        first = {}
        for p, rid in enumerate(lab):
            if rid >= 0:
                if int(rid) not in first:
                    first[int(rid)] = p
                    size[p] = self.sizes[int(rid)]
                par[p] = tile * S + w + first[int(rid)]
        self._border_lab, self._canon = lab, first
        rec = np.zeros(lay["nbytes"], np.uint8)
        rec[lay["val"][0]:lay["val"][0] + 4 * w] = val.view(np.uint8)
        rec[lay["par"][0]:lay["par"][0] + 8 * w] = par.view(np.uint8)
        rec[lay["size"][0]:lay["size"][0] + 8 * w] = size.view(np.uint8)
        return torch.from_numpy(rec)

    def border_merge(self, records, n_tiles, tile, max_diff):
        """what k_spk_border_union / _accum / _patch do, on the host: the concatenated records are the union-find forest"""
        from ros_gpu_stereo_processor_amd import rowsplit as RS
        w = self.tile.shape[1]
        lay = RS.border_record_layout(w)
        S, n2 = lay["nbytes"] // 4, 2 * w
        nd = records.numpy().copy().view(np.int32)                       # int32 view: node g = t * S + w + p, its size at g + 2w
        val = records.numpy().reshape(n_tiles, lay["nbytes"])[:, :4 * w].copy().view(np.int16).astype(np.int64)

        def find(i):
            while nd[i] != i:
                nd[i] = nd[nd[i]]
                i = nd[i]
            return int(i)
        for b in range(n_tiles - 1):
            for x in range(w):
                vb, vt = val[b][w + x], val[b + 1][x]
                if vb != self.new_val and vt != self.new_val and abs(vb - vt) <= max_diff:
                    ra, rb = find(b * S + w + w + x), find((b + 1) * S + w + x)
                    if ra != rb:
                        nd[max(ra, rb)] = min(ra, rb)
        for t in range(n_tiles):
            for p in range(n2):
                g = t * S + w + p
                if nd[g + n2] != 0:
                    r = find(g)
                    if r != g:
                        nd[r + n2] += nd[g + n2]
        for rid, p in self._canon.items():
            self.sizes[rid] = int(nd[find(tile * S + w + p) + n2])

    def apply(self, max_size):
        a = self.tile.numpy()
        for rid, n in self.sizes.items():
            if n <= max_size:
                a[self.lab == rid] = self.new_val
        return self.tile


def _speckle_frame():
    """valid background with speckles of known sizes, several of them straddling the tile borders of a 3-way split (rows 14, 28):
    a 5 x 3 blob across row 14 (15 px: removed at max_size 16 although each half is smaller AND bigger cases), a tall 1-px snake through
    two borders (30 px: kept), a blob that only touches a border row."""
    rng = np.random.default_rng(4)
    h, w = 42, 40
    a = np.full((h, w), -16, np.int16)
    a[2:40, 2:38] = (rng.integers(0, 2, (38, 36)) * 640 + 160).astype(np.int16)      # two-level noise: many small components
    a[12:17, 5:8] = 3000                                                                # 15 px across the first border
    a[6:36, 20] = 5000                                                                  # 30 px snake across both borders
    a[26:28, 30:34] = 7000                                                              # 8 px ending ON the last row of tile 1
    a[28:30, 30:32] = 7016                                                              # 4 px, connected to it through the border (|diff| 16)
    return a


def _speckle_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ros_gpu_stereo_processor_amd import rowsplit as RS
    full = _speckle_frame()
    ranges = RS.plan(full.shape[0], world)
    y0, y1 = ranges[rank]
    tile = torch.from_numpy(full[y0:y1].copy())
    out = RS.distributed_speckle(tile, rank, world, -16, 16, 32, labeler=_CpuLabels)
    q.put((rank, y0, y1, out.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_distributed_speckle_world3(oracle):
    """Row-tile split, speckle filter: every rank labels its own rows, one all-gather of the border records, merge of the components
    that touch across a border, sizes patched, then the local erase -- stitched result == cv::filterSpeckles on the whole frame."""
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_speckle_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = _speckle_frame()
    got = np.concatenate([r[3] for r in res], 0)
    exp = oracle.filter_speckles(full.copy(), -16, 16, 32)
    assert np.array_equal(got, exp)
    assert (exp[12:17, 5:8] == -16).all() and (exp[6:36, 20] == 5000).all() and (exp[26:28, 30:34] == -16).all()   # 15 px gone, 30 px kept, 8+4 px gone
    # and per-tile filtering WITHOUT the exchange would be wrong (the test frame really exercises the merge)
    naive = np.concatenate([oracle.filter_speckles(full[a:b].copy(), -16, 16, 32) for a, b in [(r[1], r[2]) for r in res]], 0)
    assert not np.array_equal(naive, exp)
