// gip_streamer.cpp -- the pipelined host link of the stereo hot path (gip_streamer_* of include/gpuimageproc_hip.h).
//
// Replaces the per-frame upload / callback-send / waitForAllStreams cycle of the reference nodelet (reference
// src/StereoProcessor.cpp:179-180 uploadMat, :284 waitForAllStreams, src/GpuSenderIfc.cpp:13-26 enqueueSend) with a ring of
// `depth` slots, each a batch of stereo pairs:
//     submit -> memcpy into the slot's pinned input (images of 4 MB and more: the right one on a helper thread)
//            -> [copy-in stream] hipMemcpyAsync H2D, event IN
//            -> [compute lane i mod L] wait IN, gip_batch_run (the whole chain, one launch set per batch; the SAD matchers of all lanes go to
//               ONE shared stream, as in bench.py), event DONE
//            -> [copy-out stream] wait DONE, hipMemcpyAsync D2H of the requested outputs into the slot's pinned output,
//               hipLaunchHostFunc(deliver), event OUT
// A slot is reused once its OUT event has completed, so upload of batch i+1, compute of batches i and i-1 and download of batch i-2 run
// concurrently and the caller blocks only when all slots are in flight.
// ADAPTIVE BATCHING (default): the caller submits pair by pair and never chooses a batch size.  A pair goes into the slot being filled; the
// slot is launched as soon as fewer than L batches are computing (the GPU would otherwise idle) or when it is full.  A slow producer
// (a 30 Hz camera) therefore gets batches of ONE pair and minimal latency; a fast one (bag replay) gets batches that grow until launch
// overhead is amortised -- a single 1242x375 pair occupies a tenth of the GPU.
// L compute lanes = L processor handles (gip_batch_run's scratch belongs to the handle): lane 0 is the caller's handle, the others are
// clones that share its rectification tables.  Delivery order = submission order: every slot's D2H + callback is enqueued on the ONE
// copy-out stream in launch order.
// A failed launch puts the streamer into a sticky error state (everything drained, the slot reset): submit / flush return that error from
// then on.  Built on the C-ABI (gip_batch_run) plus the HIP runtime; no CPU fallback.
#include <hip/hip_runtime.h>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "gip_handle_view.h"
#include "gpuimageproc_hip.h"

namespace {

struct Slot;
struct Delivery {   // argument of the host callback: one per slot
    Slot* slot;
};

struct Slot {
    gip_streamer_s* owner = nullptr;
    int index = 0;
    uint8_t* h_in[2] = {nullptr, nullptr};     // pinned, `capacity` images each
    uint8_t* d_in[2] = {nullptr, nullptr};
    int16_t* d_d16 = nullptr;
    float* d_f32 = nullptr;
    uint8_t* d_pc = nullptr;
    int16_t* h_d16 = nullptr;                   // pinned
    float* h_f32 = nullptr;
    uint8_t* h_pc = nullptr;
    hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_out = nullptr;
    bool in_flight = false;
    bool computing = false;                     // launched and ev_done not yet seen complete
    int count = 0;                              // pairs in this batch
    uint64_t first_seq = 0;
    std::vector<gip_header> headers;
    std::vector<const void*> src[2];            // inputs_pinned: the caller's images of this batch
    Delivery delivery;
};

struct Lane {
    gip_handle h = nullptr;
    bool owned = false;
    hipStream_t st = nullptr;
};

// one-slot work queue for the right image's staging copy
struct CopyHelper {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    const uint8_t* src = nullptr;
    uint8_t* dst = nullptr;
    size_t row = 0, step = 0;
    int rows = 0;
    bool busy = false, quit = false;
};

// Fill of the pinned ring.  The ring (hundreds of megabytes) is written once and read by the DMA engine, never by this core: streaming
// (non-temporal) stores skip the read-for-ownership of every destination line that memcpy's cached stores pay below glibc's own
// non-temporal threshold -- an image of the metric's size (466 KB) is below it.  SSE2 is baseline x86-64.
void stream_copy(uint8_t* dst, const uint8_t* src, size_t n)
{
#if defined(__SSE2__)
    if (n >= 4096) {
        const size_t head = (16 - ((uintptr_t)dst & 15)) & 15;
        std::memcpy(dst, src, head);
        dst += head; src += head; n -= head;
        const size_t blocks = n / 64;
        for (size_t i = 0; i < blocks; i++) {
            const __m128i a = _mm_loadu_si128((const __m128i*)(src + 64 * i)), b = _mm_loadu_si128((const __m128i*)(src + 64 * i + 16));
            const __m128i c = _mm_loadu_si128((const __m128i*)(src + 64 * i + 32)), d = _mm_loadu_si128((const __m128i*)(src + 64 * i + 48));
            _mm_stream_si128((__m128i*)(dst + 64 * i), a);
            _mm_stream_si128((__m128i*)(dst + 64 * i + 16), b);
            _mm_stream_si128((__m128i*)(dst + 64 * i + 32), c);
            _mm_stream_si128((__m128i*)(dst + 64 * i + 48), d);
        }
        std::memcpy(dst + 64 * blocks, src + 64 * blocks, n - 64 * blocks);
        _mm_sfence();      // the H2D copy is enqueued by this thread after the fill: make the streamed lines globally visible first
        return;
    }
#endif
    std::memcpy(dst, src, n);
}

void copy_rows(uint8_t* dst, const uint8_t* src, size_t row, size_t step, int rows)
{
    if (step == row) stream_copy(dst, src, row * (size_t)rows);
    else
        for (int y = 0; y < rows; y++) stream_copy(dst + (size_t)y * row, src + (size_t)y * step, row);
}

void helper_main(CopyHelper* c)
{
    std::unique_lock<std::mutex> lk(c->mu);
    for (;;) {
        c->cv.wait(lk, [&] { return c->quit || c->busy; });
        if (c->quit) return;
        lk.unlock();
        copy_rows(c->dst, c->src, c->row, c->step, c->rows);
        lk.lock();
        c->busy = false;
        c->cv.notify_all();
    }
}

}  // namespace

struct gip_streamer_s {
    gip_handle h = nullptr;
    gip_streamer_config cfg{};
    gip_streamer_cb cb = nullptr;
    void* user = nullptr;
    int device = 0;
    int capacity = 1;            // pairs per slot
    bool adaptive = true;
    hipStream_t s_in = nullptr, s_out = nullptr, s_match = nullptr;
    std::vector<Lane> lanes;
    std::vector<Slot> slots;
    int cur = 0;                 // slot being filled
    uint64_t next_seq = 0, launches = 0;
    std::atomic<uint64_t> delivered{0};
    size_t in_bytes = 0;         // one raw image
    size_t npx = 0;
    int failed = GIP_OK;         // sticky
    std::string err;
    CopyHelper helper;
};

namespace {

#define S_HIP(s, expr)                                                                       \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            (s)->err = std::string(#expr) + " failed: " + hipGetErrorString(_e);             \
            return GIP_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

void deliver(void* p)
{
    Slot* sl = ((Delivery*)p)->slot;
    gip_streamer_s* s = sl->owner;
    const size_t npx = s->npx;
    for (int i = 0; i < sl->count; i++) {
        if (s->cb) {
            gip_streamer_result r{};
            r.seq = sl->first_seq + (uint64_t)i;
            r.header = sl->headers[i];
            r.width = s->cfg.width;
            r.height = s->cfg.height;
            r.disparity16 = sl->h_d16 ? sl->h_d16 + npx * i : nullptr;
            r.disparity_f32 = sl->h_f32 ? sl->h_f32 + npx * i : nullptr;
            r.pointcloud = sl->h_pc ? sl->h_pc + npx * 32 * i : nullptr;
            s->cb(&r, s->user);
        }
        s->delivered.fetch_add(1, std::memory_order_release);
    }
}

int enqueue_slot(gip_streamer_s* s, Slot& sl)
{
    const int B = sl.count;
    const size_t npx = s->npx;
    Lane& lane = s->lanes[(size_t)(s->launches % s->lanes.size())];
    S_HIP(s, hipSetDevice(s->device));
    // copy-in
    if (s->cfg.inputs_pinned) {
        for (int e = 0; e < 2; e++)
            for (int i = 0; i < B; i++) S_HIP(s, hipMemcpyAsync(sl.d_in[e] + s->in_bytes * i, sl.src[e][i], s->in_bytes, hipMemcpyHostToDevice, s->s_in));
    } else {
        for (int e = 0; e < 2; e++) S_HIP(s, hipMemcpyAsync(sl.d_in[e], sl.h_in[e], s->in_bytes * B, hipMemcpyHostToDevice, s->s_in));
    }
    S_HIP(s, hipEventRecord(sl.ev_in, s->s_in));
    // compute: the whole chain on the batch, on this batch's lane
    S_HIP(s, hipStreamWaitEvent(lane.st, sl.ev_in, 0));
    gip_batch_desc d{};
    d.width = s->cfg.width;
    d.height = s->cfg.height;
    d.batch = B;
    d.encoding = s->cfg.encoding;
    d.outputs = s->cfg.outputs;
    d.left_raw = sl.d_in[0];
    d.right_raw = sl.d_in[1];
    d.disparity = sl.d_d16;
    d.disparity_f32 = sl.d_f32;
    d.pointcloud = sl.d_pc;
    d.matcher_stream = s->lanes.size() > 1 ? (gip_stream)s->s_match : nullptr;
    d.batches_in_flight = (int32_t)s->lanes.size();
    int rc = gip_batch_run(lane.h, &d, (gip_stream)lane.st);
    if (rc != GIP_OK) {
        s->err = std::string("gip_batch_run: ") + gip_last_error(lane.h);
        return rc;
    }
    S_HIP(s, hipEventRecord(sl.ev_done, lane.st));
    // copy-out + delivery
    S_HIP(s, hipStreamWaitEvent(s->s_out, sl.ev_done, 0));
    if (sl.h_d16) S_HIP(s, hipMemcpyAsync(sl.h_d16, sl.d_d16, npx * 2 * B, hipMemcpyDeviceToHost, s->s_out));
    if (sl.h_f32) S_HIP(s, hipMemcpyAsync(sl.h_f32, sl.d_f32, npx * 4 * B, hipMemcpyDeviceToHost, s->s_out));
    if (sl.h_pc) S_HIP(s, hipMemcpyAsync(sl.h_pc, sl.d_pc, npx * 32 * B, hipMemcpyDeviceToHost, s->s_out));
    S_HIP(s, hipLaunchHostFunc(s->s_out, deliver, &sl.delivery));
    S_HIP(s, hipEventRecord(sl.ev_out, s->s_out));
    return GIP_OK;
}

// drain everything, forget what was queued: the state after a failed launch
void quiesce(gip_streamer_s* s)
{
    (void)hipSetDevice(s->device);
    if (s->s_in) (void)hipStreamSynchronize(s->s_in);
    for (Lane& l : s->lanes)
        if (l.st) (void)hipStreamSynchronize(l.st);
    if (s->s_match) (void)hipStreamSynchronize(s->s_match);
    if (s->s_out) (void)hipStreamSynchronize(s->s_out);
    for (Slot& x : s->slots) {
        x.in_flight = false;
        x.computing = false;
        x.count = 0;
    }
}

int launch_slot(gip_streamer_s* s, Slot& sl)
{
    if (sl.count == 0) return GIP_OK;
    const int rc = enqueue_slot(s, sl);
    if (rc != GIP_OK) {
        // H2D copies or a host function of this slot may already be enqueued and the OUT event is not recorded: nothing of this batch can be
        // trusted.  Sticky error; the pairs of this batch (and of any batch still queued) are lost.
        s->failed = rc;
        quiesce(s);
        return rc;
    }
    sl.in_flight = true;
    sl.computing = true;
    s->launches++;
    s->cur = (s->cur + 1) % (int)s->slots.size();
    return GIP_OK;
}

int wait_slot(gip_streamer_s* s, Slot& sl)
{
    if (!sl.in_flight) return GIP_OK;
    S_HIP(s, hipEventSynchronize(sl.ev_out));
    sl.in_flight = false;
    sl.computing = false;
    sl.count = 0;
    return GIP_OK;
}

// batches whose compute stage has not finished yet
int computing_now(gip_streamer_s* s)
{
    int n = 0;
    for (Slot& x : s->slots) {
        if (!x.computing) continue;
        if (hipEventQuery(x.ev_done) == hipSuccess) x.computing = false;
        else n++;
    }
    (void)hipGetLastError();     // hipErrorNotReady is not an error
    return n;
}

// a batch whose results are still on their way to the host (copy-out + delivery not finished)
bool copy_out_busy(gip_streamer_s* s)
{
    bool busy = false;
    for (Slot& x : s->slots)
        if (x.in_flight && hipEventQuery(x.ev_out) != hipSuccess) busy = true;
    (void)hipGetLastError();     // hipErrorNotReady is not an error
    return busy;
}

void free_all(gip_streamer_s* s)
{
    if (s->helper.th.joinable()) {
        {
            std::lock_guard<std::mutex> lk(s->helper.mu);
            s->helper.quit = true;
        }
        s->helper.cv.notify_all();
        s->helper.th.join();
    }
    for (Slot& sl : s->slots) {
        for (int e = 0; e < 2; e++) {
            if (sl.h_in[e]) (void)hipHostFree(sl.h_in[e]);
            if (sl.d_in[e]) (void)hipFree(sl.d_in[e]);
        }
        if (sl.d_d16) (void)hipFree(sl.d_d16);
        if (sl.d_f32) (void)hipFree(sl.d_f32);
        if (sl.d_pc) (void)hipFree(sl.d_pc);
        if (sl.h_d16) (void)hipHostFree(sl.h_d16);
        if (sl.h_f32) (void)hipHostFree(sl.h_f32);
        if (sl.h_pc) (void)hipHostFree(sl.h_pc);
        if (sl.ev_in) (void)hipEventDestroy(sl.ev_in);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
        if (sl.ev_out) (void)hipEventDestroy(sl.ev_out);
    }
    for (Lane& l : s->lanes) {
        if (l.st) (void)hipStreamDestroy(l.st);
        if (l.owned && l.h) (void)gip_destroy(l.h);
    }
    if (s->s_in) (void)hipStreamDestroy(s->s_in);
    if (s->s_match) (void)hipStreamDestroy(s->s_match);
    if (s->s_out) (void)hipStreamDestroy(s->s_out);
}

}  // namespace

extern "C" {

int gip_streamer_open(gip_handle h, const gip_streamer_config* cfg, gip_streamer_cb cb, void* user, gip_streamer* out)
{
    if (!h || !cfg || !out) return GIP_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->encoding < 0 || cfg->encoding > 6 || cfg->depth < 0 || cfg->frames_per_batch < 0 || cfg->batching < 0 ||
        cfg->batching > 1 || cfg->compute_lanes < 0 || cfg->compute_lanes > 4)
        return GIP_ERR_INVALID_ARG;
    if (!(cfg->outputs & (GIP_TOPIC_DISPARITY | GIP_TOPIC_POINTCLOUD)) && !cfg->want_disparity16) return GIP_ERR_INVALID_ARG;
    if (!gip_is_stereo_model_initialised(h)) return GIP_ERR_NOT_INITIALISED;
    int mw = 0, mh = 0;
    if (gip_model_size(h, &mw, &mh) != GIP_OK || mw != cfg->width || mh != cfg->height) return GIP_ERR_INVALID_ARG;
    gip_streamer_s* s = new (std::nothrow) gip_streamer_s();
    if (!s) return GIP_ERR_HIP;
    s->h = h;
    s->cfg = *cfg;
    s->cb = cb;
    s->user = user;
    s->adaptive = cfg->batching == 0;
    const int cn = (cfg->encoding == 1 || cfg->encoding == 2) ? 3 : 1;
    s->npx = (size_t)cfg->width * cfg->height;
    s->in_bytes = s->npx * cn;
    const bool want_f32 = (cfg->outputs & GIP_TOPIC_DISPARITY) != 0, want_pc = (cfg->outputs & GIP_TOPIC_POINTCLOUD) != 0;
    // slot capacity: given, or sized so that one slot's largest buffer stays near 256 MB (64 pairs of 1242x375 without the cloud, 16 with it)
    if (cfg->frames_per_batch > 0) s->capacity = cfg->frames_per_batch;
    else {
        const size_t per_pair = s->npx * (want_pc ? 32 : 4);
        size_t c = ((size_t)256 << 20) / (per_pair ? per_pair : 1);
        s->capacity = (int)(c < 1 ? 1 : (c > 64 ? 64 : c));
        if (!s->adaptive) s->capacity = 1;         // fixed batching with no size given: one pair per launch, as before
    }
    // One lane by default: at the sizes measured the streamer is bound by the host link and by the submitting thread's copy, not by compute
    // (1242x375: 25 k pairs/s pair by pair with one lane; a second lane brought nothing).  More lanes for chains that are compute-bound.
    const int nlanes = cfg->compute_lanes > 0 ? cfg->compute_lanes : 1;
    int depth = cfg->depth == 0 ? nlanes + 2 : cfg->depth;
    if (depth < 2) depth = 2;
    s->cfg.depth = depth;
    s->cfg.frames_per_batch = s->capacity;
    const int B = s->capacity;
    auto fail = [&](int code) { free_all(s); delete s; return code; };
    s->device = gip_device(h);
    if (hipSetDevice(s->device) != hipSuccess) return fail(GIP_ERR_HIP);
    // Copy-in, compute lane(s), copy-out (and the lanes' shared matcher stream) must run side by side: with the two copy streams on one hardware
    // queue H2D and D2H no longer overlap (27.4 k -> 18.1 k pairs/s at 64 pairs per launch, same box, same library otherwise -- and which queue a
    // stream gets depends on every stream the process created before).  So the set is measured, not assumed (gip_concurrent_streams_create); beyond
    // the runtime's four hardware queues the surplus streams are ordinary ones -- the matcher stream first in line, then the copies, then the lanes.
    {
        const int nst = 2 + nlanes + (nlanes > 1 ? 1 : 0);
        std::vector<gip_stream> set(nst, nullptr);
        int rc = gip_concurrent_streams_create(s->device, nst, set.data(), nullptr);
        if (rc != GIP_OK) return fail(rc);
        int k = 0;
        if (nlanes > 1) s->s_match = (hipStream_t)set[k++];
        s->s_in = (hipStream_t)set[k++];
        s->s_out = (hipStream_t)set[k++];
        s->lanes.resize(nlanes);
        for (int i = 0; i < nlanes; i++) s->lanes[i].st = (hipStream_t)set[k++];
    }
    for (int i = 0; i < nlanes; i++) {
        Lane& l = s->lanes[i];
        if (i == 0) l.h = h;
        else {
            int rc = gip_handle_clone(h, &l.h);
            if (rc != GIP_OK) return fail(rc);
            l.owned = true;
        }
    }
    s->slots.resize(depth);
    int idx = 0;
    for (Slot& sl : s->slots) {
        sl.owner = s;
        sl.index = idx++;
        sl.delivery.slot = &sl;
        sl.headers.resize(B);
        sl.src[0].resize(B);
        sl.src[1].resize(B);
        bool ok = true;
        for (int e = 0; e < 2; e++) {
            if (!cfg->inputs_pinned) ok = ok && hipHostMalloc((void**)&sl.h_in[e], s->in_bytes * B, hipHostMallocDefault) == hipSuccess;
            ok = ok && hipMalloc((void**)&sl.d_in[e], s->in_bytes * B) == hipSuccess;
        }
        ok = ok && hipMalloc((void**)&sl.d_d16, s->npx * 2 * B) == hipSuccess;
        if (want_f32 || want_pc) ok = ok && hipMalloc((void**)&sl.d_f32, s->npx * 4 * B) == hipSuccess;
        if (want_pc) ok = ok && hipMalloc((void**)&sl.d_pc, s->npx * 32 * B) == hipSuccess;
        if (cfg->want_disparity16) ok = ok && hipHostMalloc((void**)&sl.h_d16, s->npx * 2 * B, hipHostMallocDefault) == hipSuccess;
        if (want_f32) ok = ok && hipHostMalloc((void**)&sl.h_f32, s->npx * 4 * B, hipHostMallocDefault) == hipSuccess;
        if (want_pc) ok = ok && hipHostMalloc((void**)&sl.h_pc, s->npx * 32 * B, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sl.ev_out, hipEventDisableTiming) == hipSuccess;
        if (!ok) return fail(GIP_ERR_HIP);
    }
    // the hand-over to the helper costs two thread wake-ups (tens of microseconds): worth it from a few megabytes per image on (a 1242x375 image is
    // copied in under 20 us; with the helper the 64-pair rate fell from 27.6 k to 16.4 k pairs/s)
    if (!cfg->inputs_pinned && s->in_bytes >= ((size_t)4 << 20)) s->helper.th = std::thread(helper_main, &s->helper);
    *out = s;
    return GIP_OK;
}

int gip_streamer_submit(gip_streamer s, const void* left_raw, const void* right_raw, size_t step, const gip_header* header)
{
    if (!s || !left_raw || !right_raw) return GIP_ERR_INVALID_ARG;
    if (s->failed != GIP_OK) return s->failed;
    const int cn = (s->cfg.encoding == 1 || s->cfg.encoding == 2) ? 3 : 1;
    const size_t row = (size_t)s->cfg.width * cn;
    if (step == 0) step = row;
    if (step < row) return GIP_ERR_INVALID_ARG;
    if (s->cfg.inputs_pinned && step != row) return GIP_ERR_INVALID_ARG;   // in-place H2D needs tight rows
    Slot& sl = s->slots[s->cur];
    if (sl.in_flight) {   // back-pressure: the ring is full
        int rc = wait_slot(s, sl);
        if (rc != GIP_OK) return rc;
    }
    if (sl.count >= s->capacity) {      // cannot happen (a full slot is launched at once); never write past the slot
        s->err = "streamer slot overrun";
        return GIP_ERR_INVALID_ARG;
    }
    if (sl.count == 0) sl.first_seq = s->next_seq;
    if (s->cfg.inputs_pinned) {
        sl.src[0][sl.count] = left_raw;
        sl.src[1][sl.count] = right_raw;
    } else if (!s->helper.th.joinable()) {
        copy_rows(sl.h_in[0] + s->in_bytes * sl.count, (const uint8_t*)left_raw, row, step, s->cfg.height);
        copy_rows(sl.h_in[1] + s->in_bytes * sl.count, (const uint8_t*)right_raw, row, step, s->cfg.height);
    } else {
        // large images: the right image on the helper thread, the left one here (one thread's memcpy runs at about 25 GB/s)
        CopyHelper& c = s->helper;
        {
            std::lock_guard<std::mutex> lk(c.mu);
            c.src = (const uint8_t*)right_raw;
            c.dst = sl.h_in[1] + s->in_bytes * sl.count;
            c.row = row;
            c.step = step;
            c.rows = s->cfg.height;
            c.busy = true;
        }
        c.cv.notify_all();
        copy_rows(sl.h_in[0] + s->in_bytes * sl.count, (const uint8_t*)left_raw, row, step, s->cfg.height);
        std::unique_lock<std::mutex> lk(c.mu);
        c.cv.wait(lk, [&] { return !c.busy; });
    }
    if (header) sl.headers[sl.count] = *header;
    else sl.headers[sl.count] = gip_header{};
    sl.count++;
    s->next_seq++;
    // launch when the slot is full -- or (adaptive) as soon as a compute lane is free AND nothing is on its way back to the host: a camera-rate
    // caller finds both idle and gets the latency of one pair; a caller that saturates the link does not (the copy-out of the previous batch is
    // the pipeline's slowest stage then), and launching what has arrived so far would only cut the batches short -- every launch is ~0.1 ms of
    // this thread's time, during which it copies nothing (18.8 k pairs/s at 32 pairs per launch against 27.4 k at 62, same run otherwise)
    if (sl.count == s->capacity || (s->adaptive && computing_now(s) < (int)s->lanes.size() && !copy_out_busy(s))) return launch_slot(s, sl);
    return GIP_OK;
}

int gip_streamer_flush(gip_streamer s)
{
    if (!s) return GIP_ERR_INVALID_ARG;
    if (s->failed != GIP_OK) return s->failed;
    Slot& sl = s->slots[s->cur];
    if (!sl.in_flight && sl.count > 0) {
        int rc = launch_slot(s, sl);
        if (rc != GIP_OK) return rc;
    }
    for (Slot& x : s->slots) {
        int rc = wait_slot(s, x);
        if (rc != GIP_OK) return rc;
    }
    return GIP_OK;
}

uint64_t gip_streamer_delivered(gip_streamer s) { return s ? s->delivered.load(std::memory_order_acquire) : 0; }

uint64_t gip_streamer_launches(gip_streamer s) { return s ? s->launches : 0; }

const char* gip_streamer_last_error(gip_streamer s) { return s ? s->err.c_str() : "null streamer"; }

int gip_streamer_close(gip_streamer s)
{
    if (!s) return GIP_ERR_INVALID_ARG;
    (void)hipSetDevice(s->device);
    if (s->failed == GIP_OK) (void)gip_streamer_flush(s);
    quiesce(s);
    free_all(s);
    delete s;
    return GIP_OK;
}

}  // extern "C"
