"""Row-tile split of ONE large stereo pair across ranks (BASELINE.json configs[4]: 4096x2160, ndisp 192, 4 GPUs).

Block matching is row-local apart from the vertical extent of its windows, so rank t owns output rows [y0, y1) and needs
`halo` extra RECTIFIED rows above and below:  halo = block radius + 1 (the XSOBEL prefilter is a 3x3 operator; for
NORMALIZED_RESPONSE it is block radius + prefilter radius + 1).  Each rank rectifies its own rows, swaps the boundary rows
with its neighbours (point-to-point over xGMI: RCCL send/recv, 2 x halo x W bytes per boundary and image), runs the matcher
on its extended tile, keeps the rows whose windows never touched the tile's artificial top/bottom border, and the disparity
tiles are gathered on rank 0, where the (global) speckle filter and the cloud kernel run.  The L/R consistency check is
row-local and needs nothing extra.

The same functions drive the single-GPU parity test (ranks simulated one after the other) and the gloo test of the exchange.
"""
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
