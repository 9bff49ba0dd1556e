// gip_streamer.cpp -- the pipelined host link of the stereo hot path (gip_streamer_* of include/gpuimageproc_hip.h).
//
// Replaces the per-frame upload / callback-send / waitForAllStreams cycle of the reference nodelet (reference
// src/StereoProcessor.cpp:179-180 uploadMat, :284 waitForAllStreams, src/GpuSenderIfc.cpp:13-26 enqueueSend) with a ring of
// `depth` slots, each a batch of stereo pairs:
//     submit -> memcpy into the slot's pinned input -> [copy-in stream] hipMemcpyAsync H2D, event IN
//            -> [compute stream] wait IN, gip_batch_run (the whole chain, one launch set per batch), event DONE
//            -> [copy-out stream] wait DONE, hipMemcpyAsync D2H of the requested outputs into the slot's pinned output,
//               hipLaunchHostFunc(deliver), event OUT
// A slot is reused once its OUT event has completed, so upload of batch i+1, compute of batch i and download of batch i-1 run
// concurrently and the caller blocks only when all slots are in flight.  Built on the public C-ABI only (gip_batch_run) plus the
// HIP runtime; no CPU fallback.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "gpuimageproc_hip.h"

namespace {

struct Slot;
struct Delivery {   // argument of the host callback: one per slot
    Slot* slot;
};

struct Slot {
    gip_streamer_s* owner = nullptr;
    uint8_t* h_in[2] = {nullptr, nullptr};     // pinned, frames_per_batch images each
    uint8_t* d_in[2] = {nullptr, nullptr};
    int16_t* d_d16 = nullptr;
    float* d_f32 = nullptr;
    uint8_t* d_pc = nullptr;
    int16_t* h_d16 = nullptr;                   // pinned
    float* h_f32 = nullptr;
    uint8_t* h_pc = nullptr;
    hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_out = nullptr;
    bool in_flight = false;
    int count = 0;                              // pairs in this batch
    uint64_t first_seq = 0;
    std::vector<gip_header> headers;
    std::vector<const void*> src[2];            // inputs_pinned: the caller's images of this batch
    Delivery delivery;
};

}  // namespace

struct gip_streamer_s {
    gip_handle h = nullptr;
    gip_streamer_config cfg{};
    gip_streamer_cb cb = nullptr;
    void* user = nullptr;
    int device = 0;
    hipStream_t s_in = nullptr, s_comp = nullptr, s_out = nullptr;
    std::vector<Slot> slots;
    int cur = 0;                 // slot being filled
    uint64_t next_seq = 0;
    std::atomic<uint64_t> delivered{0};
    size_t in_bytes = 0;         // one raw image
    size_t npx = 0;
    std::string err;
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

int launch_slot(gip_streamer_s* s, Slot& sl)
{
    if (sl.count == 0) return GIP_OK;
    const int B = sl.count;
    const size_t npx = s->npx;
    S_HIP(s, hipSetDevice(s->device));
    // copy-in
    if (s->cfg.inputs_pinned) {
        for (int e = 0; e < 2; e++)
            for (int i = 0; i < B; i++) S_HIP(s, hipMemcpyAsync(sl.d_in[e] + s->in_bytes * i, sl.src[e][i], s->in_bytes, hipMemcpyHostToDevice, s->s_in));
    } else {
        for (int e = 0; e < 2; e++) S_HIP(s, hipMemcpyAsync(sl.d_in[e], sl.h_in[e], s->in_bytes * B, hipMemcpyHostToDevice, s->s_in));
    }
    S_HIP(s, hipEventRecord(sl.ev_in, s->s_in));
    // compute: the whole chain on the batch
    S_HIP(s, hipStreamWaitEvent(s->s_comp, sl.ev_in, 0));
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
    int rc = gip_batch_run(s->h, &d, (gip_stream)s->s_comp);
    if (rc != GIP_OK) {
        s->err = std::string("gip_batch_run: ") + gip_last_error(s->h);
        return rc;
    }
    S_HIP(s, hipEventRecord(sl.ev_done, s->s_comp));
    // copy-out + delivery
    S_HIP(s, hipStreamWaitEvent(s->s_out, sl.ev_done, 0));
    if (sl.h_d16) S_HIP(s, hipMemcpyAsync(sl.h_d16, sl.d_d16, npx * 2 * B, hipMemcpyDeviceToHost, s->s_out));
    if (sl.h_f32) S_HIP(s, hipMemcpyAsync(sl.h_f32, sl.d_f32, npx * 4 * B, hipMemcpyDeviceToHost, s->s_out));
    if (sl.h_pc) S_HIP(s, hipMemcpyAsync(sl.h_pc, sl.d_pc, npx * 32 * B, hipMemcpyDeviceToHost, s->s_out));
    S_HIP(s, hipLaunchHostFunc(s->s_out, deliver, &sl.delivery));
    S_HIP(s, hipEventRecord(sl.ev_out, s->s_out));
    sl.in_flight = true;
    return GIP_OK;
}

int wait_slot(gip_streamer_s* s, Slot& sl)
{
    if (!sl.in_flight) return GIP_OK;
    S_HIP(s, hipEventSynchronize(sl.ev_out));
    sl.in_flight = false;
    sl.count = 0;
    return GIP_OK;
}

void free_all(gip_streamer_s* s)
{
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
    if (s->s_in) (void)hipStreamDestroy(s->s_in);
    if (s->s_comp) (void)hipStreamDestroy(s->s_comp);
    if (s->s_out) (void)hipStreamDestroy(s->s_out);
}

}  // namespace

extern "C" {

int gip_streamer_open(gip_handle h, const gip_streamer_config* cfg, gip_streamer_cb cb, void* user, gip_streamer* out)
{
    if (!h || !cfg || !out) return GIP_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->encoding < 0 || cfg->encoding > 6 || cfg->depth < 0 || cfg->frames_per_batch < 0) return GIP_ERR_INVALID_ARG;
    if (!(cfg->outputs & (GIP_TOPIC_DISPARITY | GIP_TOPIC_POINTCLOUD)) && !cfg->want_disparity16) return GIP_ERR_INVALID_ARG;
    if (!gip_is_stereo_model_initialised(h)) return GIP_ERR_NOT_INITIALISED;
    int mw = 0, mh = 0;
    if (gip_model_size(h, &mw, &mh) != GIP_OK || mw != cfg->width || mh != cfg->height) return GIP_ERR_INVALID_ARG;
    gip_streamer_s* s = new (std::nothrow) gip_streamer_s();
    if (!s) return GIP_ERR_HIP;
    s->h = h;
    s->cfg = *cfg;
    if (s->cfg.depth == 0) s->cfg.depth = 3;
    if (s->cfg.depth < 2) s->cfg.depth = 2;
    if (s->cfg.frames_per_batch == 0) s->cfg.frames_per_batch = 1;
    s->cb = cb;
    s->user = user;
    const int cn = (cfg->encoding == 1 || cfg->encoding == 2) ? 3 : 1;
    s->npx = (size_t)cfg->width * cfg->height;
    s->in_bytes = s->npx * cn;
    const int B = s->cfg.frames_per_batch;
    auto fail = [&](int code) { free_all(s); delete s; return code; };
    s->device = gip_device(h);
    if (hipSetDevice(s->device) != hipSuccess) return fail(GIP_ERR_HIP);
    if (hipStreamCreateWithFlags(&s->s_in, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&s->s_comp, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&s->s_out, hipStreamNonBlocking) != hipSuccess)
        return fail(GIP_ERR_HIP);
    s->slots.resize(s->cfg.depth);
    const bool want_f32 = (cfg->outputs & GIP_TOPIC_DISPARITY) != 0, want_pc = (cfg->outputs & GIP_TOPIC_POINTCLOUD) != 0;
    for (Slot& sl : s->slots) {
        sl.owner = s;
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
    *out = s;
    return GIP_OK;
}

int gip_streamer_submit(gip_streamer s, const void* left_raw, const void* right_raw, size_t step, const gip_header* header)
{
    if (!s || !left_raw || !right_raw) return GIP_ERR_INVALID_ARG;
    const int cn = (s->cfg.encoding == 1 || s->cfg.encoding == 2) ? 3 : 1;
    const size_t row = (size_t)s->cfg.width * cn;
    if (step == 0) step = row;
    if (step < row) return GIP_ERR_INVALID_ARG;
    Slot& sl = s->slots[s->cur];
    if (sl.in_flight) {   // back-pressure: the ring is full
        int rc = wait_slot(s, sl);
        if (rc != GIP_OK) return rc;
    }
    if (sl.count == 0) sl.first_seq = s->next_seq;
    const void* src[2] = {left_raw, right_raw};
    if (s->cfg.inputs_pinned && step != row) return GIP_ERR_INVALID_ARG;   // in-place H2D needs tight rows
    for (int e = 0; e < 2; e++) {
        if (s->cfg.inputs_pinned) {
            sl.src[e][sl.count] = src[e];
            continue;
        }
        uint8_t* dst = sl.h_in[e] + s->in_bytes * sl.count;
        if (step == row) std::memcpy(dst, src[e], s->in_bytes);
        else
            for (int y = 0; y < s->cfg.height; y++) std::memcpy(dst + (size_t)y * row, (const uint8_t*)src[e] + (size_t)y * step, row);
    }
    if (header) sl.headers[sl.count] = *header;
    else sl.headers[sl.count] = gip_header{};
    sl.count++;
    s->next_seq++;
    if (sl.count == s->cfg.frames_per_batch) {
        int rc = launch_slot(s, sl);
        if (rc != GIP_OK) return rc;
        s->cur = (s->cur + 1) % (int)s->slots.size();
    }
    return GIP_OK;
}

int gip_streamer_flush(gip_streamer s)
{
    if (!s) return GIP_ERR_INVALID_ARG;
    Slot& sl = s->slots[s->cur];
    if (!sl.in_flight && sl.count > 0) {
        int rc = launch_slot(s, sl);
        if (rc != GIP_OK) return rc;
        s->cur = (s->cur + 1) % (int)s->slots.size();
    }
    for (Slot& x : s->slots) {
        int rc = wait_slot(s, x);
        if (rc != GIP_OK) return rc;
    }
    return GIP_OK;
}

uint64_t gip_streamer_delivered(gip_streamer s) { return s ? s->delivered.load(std::memory_order_acquire) : 0; }

const char* gip_streamer_last_error(gip_streamer s) { return s ? s->err.c_str() : "null streamer"; }

int gip_streamer_close(gip_streamer s)
{
    if (!s) return GIP_ERR_INVALID_ARG;
    (void)hipSetDevice(s->device);
    (void)gip_streamer_flush(s);
    if (s->s_in) (void)hipStreamSynchronize(s->s_in);
    if (s->s_comp) (void)hipStreamSynchronize(s->s_comp);
    if (s->s_out) (void)hipStreamSynchronize(s->s_out);
    free_all(s);
    delete s;
    return GIP_OK;
}

}  // extern "C"
