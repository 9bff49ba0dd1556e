"""Row-tile split of ONE large stereo pair across ranks (BASELINE.json configs[4]: 4096x2160, ndisp 192, 4 GPUs).

Block matching is row-local apart from the vertical extent of its windows, so rank t owns output rows [y0, y1) and needs
`halo` extra RECTIFIED rows above and below:  halo = block radius + 1 (the XSOBEL prefilter is a 3x3 operator; for
NORMALIZED_RESPONSE it is block radius + prefilter radius + 1).  Each rank rectifies its own rows, swaps the boundary rows
with its neighbours (point-to-point over xGMI: RCCL send/recv, 2 x halo x W bytes per boundary and image), runs the matcher
on its extended tile and keeps the rows whose windows never touched the tile's artificial top/bottom border.  The L/R
consistency check is row-local and needs nothing extra.  The speckle filter is global (a component may cross tile borders):
every rank labels the components of its own rows, the ranks all-gather a 20-byte-per-column record of their first and last rows, every
rank merges the components that touch across a border ON THE DEVICE, patches the sizes of its own border components to the size of the
whole component and erases what is small (distributed_speckle below; the single exchange of SURVEY.md section 8(e)).  Cloud and float
disparity are produced per tile.

Two forms of the same split: the functions below are the one-process-per-GPU form (torch.distributed: RCCL point-to-point for the halo
rows, one all-gather for the border records); class RowSplit at the end holds the native form (gip_rowsplit_* of the C-ABI: one process
owns N handles, everything preallocated, hipMemcpyPeerAsync).  The same functions drive the single-GPU parity test (ranks simulated one
after the other) and the gloo test of the exchange.
"""
import numpy as np
import torch
import torch.distributed as dist


def halo_rows(block_size, prefilter_type=1, prefilter_size=9):
    r = block_size // 2
    return r + 1 if prefilter_type == 1 else r + prefilter_size // 2 + 1


def plan(height, world_size):
    """Output row range [y0, y1) of every rank (contiguous, near-equal)."""
    base, rem = divmod(height, world_size)
    out, y = [], 0
    for t in range(world_size):
        n = base + (1 if t < rem else 0)
        out.append((y, y + n))
        y += n
    return out


def extended_range(height, y0, y1, halo):
    """Rows of the full frame a tile must see: its own rows plus the halo, clipped to the frame.  The first row is kept
    EVEN: the XSOBEL prefilter forces the last row of an odd-height image to the cap value, so the bottom tile must have the
    parity of the full frame (interior rows carry no parity dependence)."""
    e0 = max(0, y0 - halo)
    return e0 - (e0 & 1), min(height, y1 + halo)


def exchange_halos(tile, rank, world_size, ranges, height, halo, group=None):
    """tile: (rows, W) tensor with this rank's own rows [y0, y1) of `ranges`.  Returns the tile extended to
    extended_range(...) with the rows received from the neighbouring ranks.  How many rows cross each boundary follows
    from the plan alone, so every rank derives its send and receive counts without negotiation."""
    def ext(t):
        return extended_range(height, ranges[t][0], ranges[t][1], halo)
    y0, y1 = ranges[rank]
    e0, e1 = ext(rank)
    ops, top, bottom = [], None, None
    if rank > 0:
        n_recv = y0 - e0                        # rows I need from above
        n_send = ext(rank - 1)[1] - y0          # rows the rank above needs from me
        assert ranges[rank - 1][1] - ranges[rank - 1][0] >= n_recv and y1 - y0 >= n_send, "tile thinner than the halo"
        top = torch.empty((n_recv,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
        ops.append(dist.P2POp(dist.isend, tile[:n_send].contiguous(), rank - 1, group))
        ops.append(dist.P2POp(dist.irecv, top, rank - 1, group))
    if rank < world_size - 1:
        n_recv = e1 - y1
        n_send = y1 - ext(rank + 1)[0]
        assert ranges[rank + 1][1] - ranges[rank + 1][0] >= n_recv and y1 - y0 >= n_send, "tile thinner than the halo"
        bottom = torch.empty((n_recv,) + tuple(tile.shape[1:]), dtype=tile.dtype, device=tile.device)
        ops.append(dist.P2POp(dist.isend, tile[tile.shape[0] - n_send:].contiguous(), rank + 1, group))
        ops.append(dist.P2POp(dist.irecv, bottom, rank + 1, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    parts = [p for p in (top, tile, bottom) if p is not None]
    return torch.cat(parts, 0)


def crop_valid(ext_out, height, y0, y1, halo):
    """Rows of the extended tile's output that belong to [y0, y1)."""
    e0, _ = extended_range(height, y0, y1, halo)
    return ext_out[y0 - e0:y1 - e0]


def match_tile(rect_l_ext, rect_r_ext, params):
    """StereoBM on an extended tile (HIP path through the C-ABI)."""
    from . import kernels as K
    return K.stereobm(rect_l_ext, rect_r_ext, params)


def gather_rows(tile_out, rank, world_size, ranges, dst=0, group=None):
    """Collect the per-rank output tiles on `dst` (returns the stitched frame there, None elsewhere).  Byte views: neither
    RCCL nor gloo has a 16-bit integer type."""
    w = tile_out.shape[1]
    if rank == dst:
        h = ranges[-1][1]
        full = torch.empty((h, w), dtype=tile_out.dtype, device=tile_out.device)
        y0, y1 = ranges[dst]
        full[y0:y1] = tile_out
        reqs = []
        for t, (a, b) in enumerate(ranges):
            if t == dst:
                continue
            buf = torch.empty((b - a, w), dtype=tile_out.dtype, device=tile_out.device)
            reqs.append((dist.irecv(buf.view(torch.uint8), t, group), buf, a, b))
        for req, buf, a, b in reqs:
            req.wait()
            full[a:b] = buf
        return full
    dist.send(tile_out.contiguous().view(torch.uint8), dst, group)
    return None


def split_stereobm_simulated(rect_l, rect_r, params, world_size, block_size, prefilter_type=1, prefilter_size=9):
    """Single-process rehearsal of the split: every 'rank' gets exactly the rows it would hold after the halo exchange."""
    h = rect_l.shape[0]
    halo = halo_rows(block_size, prefilter_type, prefilter_size)
    ranges = plan(h, world_size)
    out = torch.empty((h, rect_l.shape[1]), dtype=torch.int16, device=rect_l.device)
    for (y0, y1) in ranges:
        e0, e1 = extended_range(h, y0, y1, halo)
        d = match_tile(rect_l[e0:e1].contiguous(), rect_r[e0:e1].contiguous(), params)
        out[y0:y1] = crop_valid(d, h, y0, y1, halo)
    return out


# ------------------------------------------------------------------------------------------------------------------ speckle
def border_record_layout(w):
    """Byte layout of a tile's border record (k_spk_border_record, csrc/k_post.hip): for the 2 w border pixels p (p < w: the tile's first
    row, else its last row)  value int16[2w] @0,  par int32[2w] @4w (union-find parent: node of the component's canonical border pixel of
    this tile; the node of pixel p of tile t is  t * nbytes / 4 + w + p  -- an int32 index into the concatenated records; an invalid pixel
    points at itself),  size int32[2w] @12w (tile-local component size at the canonical pixel, 0 elsewhere); padded to a multiple of 256 bytes."""
    return dict(val=(0, 2 * w), par=(4 * w, 2 * w), size=(12 * w, 2 * w), nbytes=(20 * w + 255) // 256 * 256)


def distributed_speckle(tile, rank, world_size, new_val, max_size, max_diff, group=None, labeler=None):
    """cv::filterSpeckles on a frame that is split into row tiles: `tile` holds this rank's rows (int16, modified in place and
    returned).  Every rank labels its own rows, publishes its border record, the ranks all-gather the records (the single exchange of
    SURVEY.md section 8(e): 20 bytes per column and rank), and every rank merges the components that cross a border and patches the sizes
    of its own border components -- ON THE DEVICE for the default labeler (kernels.SpeckleLabels: gipk_speckle_border_record / _merge),
    no host round trip.  labeler(tile, new_val, max_diff) -> object with border_record() -> uint8 tensor, border_merge(records, n_tiles,
    tile, max_diff), apply(max_size); border_record takes the rank."""
    if labeler is None:
        from .kernels import SpeckleLabels as labeler
    lab = labeler(tile, new_val, max_diff)
    if world_size > 1:
        rec = lab.border_record(rank)
        everyone = [torch.empty_like(rec) for _ in range(world_size)]
        dist.all_gather(everyone, rec, group=group)
        lab.border_merge(torch.cat(everyone), world_size, rank, max_diff)
    return lab.apply(max_size)


def split_speckle_simulated(disp, world_size, new_val, max_size, max_diff):
    """Single-process rehearsal of distributed_speckle on the HIP labeler: every 'rank' labels its own rows, the border records are
    concatenated as the all-gather would, every tile merges on the device, patches its sizes, erases; tiles stitched."""
    from .kernels import SpeckleLabels
    h = disp.shape[0]
    ranges = plan(h, world_size)
    tiles = [disp[a:b].contiguous().clone() for a, b in ranges]
    labs = [SpeckleLabels(t, new_val, max_diff) for t in tiles]
    records = torch.cat([lab.border_record(t) for t, lab in enumerate(labs)])
    for t, lab in enumerate(labs):
        if world_size > 1:
            lab.border_merge(records.clone(), world_size, t, max_diff)      # every rank merges on its own copy
        lab.apply(max_size)
    return torch.cat(tiles, 0)


def rowsplit_frame(raw_l, raw_r, maps, params, rank, world_size, speckle_size, speckle_diff16, Q=None, cx_l=0.0, cx_r=0.0, group=None, want_cloud=True):
    """One frame of BASELINE configs[4] on this rank: rectify the rank's own rows of both eyes (the raw frame is resident on every
    rank), swap halo rows with the neighbours, match the extended tile (StereoBM incl. L/R check), keep the rank's rows, run the
    distributed speckle filter, and produce float disparity + cloud for the rank's rows.  Returns (y0, y1, disparity int16 tile,
    float tile or None, cloud tile or None).  maps = [map1_l, map2_l, map1_r, map2_r] full-frame device tensors."""
    from . import kernels as K
    h, w = raw_l.shape
    halo = halo_rows(params.block_size, params.prefilter_type, params.prefilter_size)
    ranges = plan(h, world_size)
    y0, y1 = ranges[rank]
    rl = K.remap(raw_l, maps[0][y0:y1], maps[1][y0:y1])
    rr = K.remap(raw_r, maps[2][y0:y1], maps[3][y0:y1])
    if world_size > 1:
        rl = exchange_halos(rl, rank, world_size, ranges, h, halo, group)
        rr = exchange_halos(rr, rank, world_size, ranges, h, halo, group)
    d = crop_valid(match_tile(rl.contiguous(), rr.contiguous(), params), h, y0, y1, halo).contiguous()
    if world_size == 1:
        d = d.clone()
    if speckle_size > 0:
        d = distributed_speckle(d, rank, world_size, (params.min_disparity - 1) * 16, speckle_size, speckle_diff16, group)
    f32 = cloud = None
    if want_cloud and Q is not None:
        # the reprojection depends on the row index of the FULL frame: the kernel adds the tile origin to its row number
        e0, _ = extended_range(h, y0, y1, halo)
        colour = rl[y0 - e0:y1 - e0].contiguous()
        f32, cloud = K.disparity_outputs(d, colour, np.array(Q, np.float64).reshape(4, 4), cx_l, cx_r, params.min_disparity, y_origin=y0)
    return y0, y1, d, f32, cloud


# ------------------------------------------------------------------------------------------------------------------ native form
class RowSplit:
    """ctypes holder of gip_rowsplit_* (include/gpuimageproc_hip.h): ONE process owns n_tiles processor handles, every buffer is
    allocated at open, halo rows travel by hipMemcpyPeerAsync and the components that cross tile borders are merged on the device.
    The functions above are the one-process-per-GPU (torch.distributed / RCCL) form of the same split."""

    def __init__(self, width, height, calib, params, n_tiles, encoding="mono8", outputs=None, max_speckle_size=0, max_speckle_diff=0.0, devices=None,
                 halo_mode=0, host_threads=0, gather_mode=0):
        import ctypes as C
        from . import capi
        from .pipeline import ENCODINGS
        from .processor import camera_info
        self._C, self._capi, self._lib = C, capi, capi.lib()
        self.w, self.h, self.n_tiles = width, height, n_tiles
        cfg = capi.RowSplitConfig()
        cfg.width, cfg.height, cfg.encoding = width, height, ENCODINGS[encoding]
        cfg.outputs = (capi.TOPIC_DISPARITY | capi.TOPIC_POINTCLOUD) if outputs is None else outputs
        cfg.n_tiles = n_tiles
        for i in range(16):
            cfg.devices[i] = devices[i] if devices is not None and i < len(devices) else -1
        cfg.halo_mode, cfg.host_threads, cfg.gather_mode = halo_mode, host_threads, gather_mode
        self.outputs = cfg.outputs
        self.channels = 3 if encoding in ("bgr8", "rgb8") else 1
        li, ri = camera_info(width, height, **calib["left"]), camera_info(width, height, **calib["right"])
        self.handle = C.c_void_p()
        rc = self._lib.gip_rowsplit_open(C.byref(li), C.byref(ri), C.byref(params), int(max_speckle_size), float(max_speckle_diff), C.byref(cfg), C.byref(self.handle))
        if rc != capi.OK:
            raise capi.GipError(rc, "gip_rowsplit_open")
        self._keep = []

    def _ck(self, rc):
        if rc != self._capi.OK:
            raise self._capi.GipError(rc, self._lib.gip_rowsplit_last_error(self.handle).decode())

    def plan(self, tile):
        C = self._C
        v = [C.c_int() for _ in range(5)]
        self._ck(self._lib.gip_rowsplit_plan(self.handle, tile, *[C.byref(x) for x in v]))
        return dict(zip(("y0", "y1", "e0", "e1", "device"), [x.value for x in v]))

    def set_inputs(self, tile, left_raw, right_raw):
        """left_raw / right_raw: contiguous uint8 CUDA tensors of the FULL raw frame on the tile's device (kept alive here)."""
        assert left_raw.is_cuda and left_raw.is_contiguous() and right_raw.is_contiguous()
        self._keep = [k for k in self._keep if k[0] != tile] + [(tile, left_raw, right_raw)]
        self._ck(self._lib.gip_rowsplit_set_inputs_device(self.handle, tile, left_raw.data_ptr(), right_raw.data_ptr()))

    def upload(self, left_raw, right_raw):
        """host numpy frames -> every tile's device"""
        l, r = np.ascontiguousarray(left_raw), np.ascontiguousarray(right_raw)
        assert l.shape[:2] == (self.h, self.w) and r.shape == l.shape and l.dtype == np.uint8
        self._ck(self._lib.gip_rowsplit_upload(self.handle, l.ctypes.data, r.ctypes.data, 0))

    def run(self):
        self._ck(self._lib.gip_rowsplit_run(self.handle))

    def wait(self):
        self._ck(self._lib.gip_rowsplit_wait(self.handle))

    def run_serial(self):
        t = (self._capi.RowSplitTiming * self.n_tiles)()
        self._ck(self._lib.gip_rowsplit_run_serial(self.handle, t))
        return [{n: getattr(x, n) for n, _ in self._capi.RowSplitTiming._fields_} for x in t]

    def last_timing(self, tile):
        t = self._capi.RowSplitTiming()
        self._ck(self._lib.gip_rowsplit_last_timing(self.handle, tile, self._C.byref(t)))
        return {n: getattr(t, n) for n, _ in self._capi.RowSplitTiming._fields_}

    def download(self, want_f32=True, want_cloud=True):
        d16 = np.empty((self.h, self.w), np.int16)
        f32 = np.empty((self.h, self.w), np.float32) if want_f32 and (self.outputs & self._capi.TOPIC_DISPARITY) else None
        pc = np.empty((self.h, self.w, 32), np.uint8) if want_cloud and (self.outputs & self._capi.TOPIC_POINTCLOUD) else None
        self._ck(self._lib.gip_rowsplit_download(self.handle, d16.ctypes.data, f32.ctypes.data if f32 is not None else None, pc.ctypes.data if pc is not None else None))
        return d16, f32, pc

    def close(self):
        if self.handle:
            self._lib.gip_rowsplit_close(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
