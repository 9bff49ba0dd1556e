// bm_common.h -- argument block shared by the block-matching kernels (k_stereobm.hip, k_stereobm_c4.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct BmArgs {
    const uint8_t* L;
    const uint8_t* R;
    size_t pitch;
    int w, h;
    int ndisp, mindisp, wsz2, lofs, rofs, width1;
    int ftzero, tex_thr, uniq;
    int row0, row1, rows_per_chunk;
    int cbeg, cend;      // fast kernel: output columns (relative index c) [cbeg, cend)
    int nleft, nright;   // generic kernel: c in [0,nleft) U [width1-nright, width1)
    int TC;              // generic kernel: columns per tile
    int16_t* disp;
    size_t dpitch;       // elements
    int32_t* cost;       // may be null; min-SAD plane (int32, or uint16 when cost16 is set)
    size_t cpitch;       // elements
    int cost16;          // column-sum matcher only: the plane is uint16 (its window sums are < 65535 by construction)
    size_t img_bs, disp_bs, cost_bs;  // batch strides in bytes
    int filtered;
    int ntiles, nchunks, nimg;   // column-sum matcher: extent of its (tile, row chunk, image) work space (set by the launcher)
    int batch;                   // column-sum matcher: images in the batch (nimg = batch, or ceil(batch / 2) when a wave takes two images)
};

// Host-side launch context of the dominant kernel (the SAD matcher); every field may be null.  Travels as an argument from
// gip_batch_run down to the launcher (no global state).
struct BmLaunchCtx {
    hipEvent_t ev0 = nullptr, ev1 = nullptr;     // recorded around the matcher kernel, on the stream it is launched on
    hipStream_t matcher_stream = nullptr;        // launch the matcher here instead of the caller's stream ...
    hipEvent_t fork = nullptr, join = nullptr;   // ... fenced by these two events (required when matcher_stream is set)
    bool co_scheduled = false;                   // kernels of other batches run next to this launch (gip_batch_desc::batches_in_flight > 1)
    bool defer_join = false;                     // forked launch: do NOT make the caller's stream wait for the matcher; the event to wait for is
    mutable hipEvent_t pending_join = nullptr;   // left here instead (gip_batch_run's FRONT / BACK phases)
};

// k_stereobm_c4.hip: column-sum matcher, every output column.  Returns GIP_OK and sets *handled when a kernel
// instantiation exists for (ndisp, block radius); otherwise leaves *handled false and launches nothing.
int gip_bm_c4_launch(const BmArgs& a, int batch, hipStream_t st, const BmLaunchCtx* ctx, bool* handled);
// true when gip_bm_c4_launch would handle these arguments (nothing is launched)
bool gip_bm_c4_can(const BmArgs& a);

// k_stereobm_wide.hip: 32-bit fallback for windows with 2*cap*block^2 >= 65535, all output columns
int gip_bm_wide_launch(const BmArgs& a, int nrows, int batch, hipStream_t st);
