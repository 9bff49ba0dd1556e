// gip_rowsplit.hip -- row-tile split of ONE large stereo pair over N processor handles in one process (gip_rowsplit_* of
// include/gpuimageproc_hip.h; BASELINE configs[4]: 4096x2160, ndisp 192, 4 GPUs, halo exchange over xGMI).
//
// What it replaces: the reference runs every stage of a frame as one launch on one GPU and then blocks (reference
// src/GPUStereoProcessor.cpp:283 block_matcher_gpu_->compute, src/StereoProcessor.cpp:284 waitForAllStreams) and filters speckles on
// the CPU with the stream drained (src/GPUStereoProcessor.cpp:356-385).  Here the frame is cut into row tiles:
//   * block matching is row-local apart from the vertical extent of its windows, so tile t owns output rows [y0, y1) and needs
//     `halo` more RECTIFIED rows on either side (block radius + 1: the XSOBEL prefilter is a 3x3 operator; + the prefilter radius for
//     NORMALIZED_RESPONSE).  The L/R check and the sub-pixel step are row-local: the tile's own rows equal the full frame's bit for bit;
//   * the speckle filter is global: every tile labels its own rows, the tiles swap a 20-byte-per-column record of their border rows
//     and merge the components that cross a border ON THE DEVICE (k_spk_border_merge, k_post.hip);
//   * everything is preallocated at open; the steady state is kernel launches, peer copies and event waits -- no allocation, no host
//     synchronisation, no host union-find.  One host thread per tile enqueues that tile's work, so the launch overhead of N tiles is paid
//     in parallel (about 25 enqueue calls per tile and frame).  (Replaying each tile's launches from hipGraphs was measured: 3-7 % SLOWER
//     per tile on ROCm 7.2 -- the graph's kernel nodes do not follow each other any closer than stream launches do -- and dropped.)
// Data layout: the rectified rows of both eyes INTERLEAVED in one buffer (row i = [left row i | right row i], pitch 2 * rp): the halo
// rows of both eyes are then ONE contiguous block per neighbour, i.e. one hipMemcpyPeerAsync per boundary and direction.
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "gip_internal.h"
#include "gip_handle_view.h"
#include "bm_common.h"

namespace {

struct Tile {
    gip_handle h = nullptr;
    GipHandleView v{};
    int t = 0, device = 0;
    int y0 = 0, y1 = 0, e0 = 0, e1 = 0;       // own rows, extended rows
    hipStream_t st = nullptr;
    const uint8_t* raw[2] = {nullptr, nullptr};    // full raw frame on this device (caller's, or `own_raw` after gip_rowsplit_upload)
    uint8_t* own_raw[2] = {nullptr, nullptr};
    uint8_t* stage[2] = {nullptr, nullptr};        // pinned staging of gip_rowsplit_upload (tile 0 only holds the host copy)
    uint8_t* mono[2] = {nullptr, nullptr};         // full-frame mono (encoding != mono8)
    uint8_t* color = nullptr;                      // full-frame BGR (colour input + cloud)
    uint8_t* rect = nullptr;                       // (e1 - e0) rows x [left rp | right rp]
    uint8_t* rect_color = nullptr;                 // own rows x w x 3 (colour input + cloud)
    void* bm_scratch = nullptr;
    size_t bm_scratch_bytes = 0;
    int16_t* d_ext = nullptr;                      // (e1 - e0) x w
    void* spk_scratch = nullptr;
    size_t spk_scratch_bytes = 0;
    uint8_t* pub_record = nullptr;                 // this tile's border record as the other tiles pull it
    uint8_t* records = nullptr;                    // the records of all n_tiles tiles: this tile's working copy (the union-find forest of its merge)
    int32_t* border_roots = nullptr;               // 2 w
    void* w_tab = nullptr;                         // gip_w_table_build: what the cloud kernel derives from the disparity value alone
    float* f32 = nullptr;                          // own rows
    uint8_t* pc = nullptr;
    hipEvent_t ev_rect = nullptr, ev_rec = nullptr, ev_done = nullptr;   // cross-tile dependencies
    hipEvent_t tev[6] = {};                        // timing: start, front, halo, match, gather, end
    hipEvent_t sev = nullptr;                      // gip_rowsplit_run_serial: start of the phase being timed
    gip_rowsplit_timing serial_tm{};
    bool serial_valid = false;
    bool gather_kernel = false;                    // collect the border records with one kernel (else one copy per tile)
    bool ran = false;
};

}  // namespace

struct gip_rowsplit_s {
    gip_rowsplit_config cfg{};
    gip_bm_params bp{};
    int spk_size = 0, spk_diff16 = 0;
    int halo = 0;
    size_t rp = 0;               // bytes of one eye's rectified row (width rounded up to 64)
    size_t rec_bytes = 0;
    std::vector<Tile> tiles;
    std::string err;
    std::mutex err_mu;
    bool frames_done = false;    // at least one frame has been enqueued (ev_done recorded on every tile)
    // enqueueing host threads (one per tile)
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    uint64_t gen = 0;
    int pending = 0;
    bool quit = false;
    std::vector<int> worker_rc;
    std::atomic<uint64_t>* rect_gen = nullptr;   // per tile: frames whose ev_rect / ev_rec have been RECORDED (host side)
    std::atomic<uint64_t>* rec_gen = nullptr;
    uint64_t frame = 0;
};

namespace {

int rs_fail(gip_rowsplit_s* rs, int code, const std::string& msg)
{
    std::lock_guard<std::mutex> lk(rs->err_mu);
    rs->err = msg;
    return code;
}

#define RS_HIP(rs, expr)                                                                        \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) return rs_fail(rs, GIP_ERR_HIP, std::string(#expr) + " failed: " + hipGetErrorString(_e)); \
    } while (0)
#define RS_STEP(rs, expr, what)                                                                 \
    do {                                                                                        \
        int _rc = (expr);                                                                       \
        if (_rc != GIP_OK) return rs_fail(rs, _rc, std::string(what) + ": " + (_rc == GIP_ERR_HIP ? gip_last_hip_error_text() : gip_status_string(_rc))); \
    } while (0)

int halo_rows(const gip_bm_params& p)
{
    const int r = p.block_size / 2;
    return p.prefilter_type == GIP_PREFILTER_XSOBEL ? r + 1 : r + p.prefilter_size / 2 + 1;
}

// Rows of the full frame a tile must see: its own rows plus the halo, clipped to the frame.  The first row is kept EVEN: the XSOBEL
// prefilter forces the last row of an odd-height image to the cap value, so the bottom tile must have the parity of the full frame
// (interior rows carry no parity dependence).
void extended_range(int height, int y0, int y1, int halo, int* e0, int* e1)
{
    int a = y0 - halo;
    if (a < 0) a = 0;
    a -= a & 1;
    int b = y1 + halo;
    if (b > height) b = height;
    *e0 = a;
    *e1 = b;
}

int raw_channels(int enc) { return (enc == 1 || enc == 2) ? 3 : 1; }

// ---------------------------------------------------------------------------------------------------------------- per-tile phases
// Every phase only enqueues work on the tile's stream.  `k` = the frame number (for the host-side "has been recorded" handshakes).

int tile_begin(gip_rowsplit_s* rs, Tile& T)
{
    RS_HIP(rs, hipSetDevice(T.device));
    // the previous frame of every OTHER tile must be complete before this tile's buffers are overwritten: the neighbours pull halo rows
    // out of `rect`, everybody pulls this tile's record (ev_done is recorded after a tile's last pull)
    if (rs->frames_done)
        for (Tile& O : rs->tiles)
            if (O.t != T.t) RS_HIP(rs, hipStreamWaitEvent(T.st, O.ev_done, 0));
    RS_HIP(rs, hipEventRecord(T.tev[0], T.st));
    return GIP_OK;
}

int front_body(gip_rowsplit_s* rs, Tile& T)
{
    const int w = rs->cfg.width, hh = rs->cfg.height, enc = rs->cfg.encoding;
    const size_t npx = (size_t)w * hh;
    const bool want_pc = (rs->cfg.outputs & GIP_TOPIC_POINTCLOUD) != 0;
    if (!T.raw[0] || !T.raw[1]) return rs_fail(rs, GIP_ERR_EMPTY_SLOT, "rowsplit: no input frame on tile " + std::to_string(T.t));
    const uint8_t* mono[2] = {T.raw[0], T.raw[1]};
    const uint8_t* color_l = nullptr;
    if (enc != 0) {
        // colour conversion of the whole frame on every tile (elementwise, 2 bytes per pixel: cheap next to gathering the source rows a
        // distorted tile touches)
        for (int i = 0; i < 2; i++) {
            if (enc == 1 || enc == 2) RS_STEP(rs, gipk_cvt_bgr_to_gray(T.raw[i], (size_t)w * 3, T.mono[i], w, w, hh, enc == 2, 1, 0, 0, T.st), "bgr->gray");
            else RS_STEP(rs, gipk_cvt_bayer_to_gray(T.raw[i], w, T.mono[i], w, w, hh, enc - 3, 1, 0, 0, T.st), "bayer->gray");
            mono[i] = T.mono[i];
        }
        if (want_pc) {
            if (enc == 1) color_l = T.raw[0];
            else {
                if (enc == 2) RS_STEP(rs, gipk_cvt_swap_rb(T.raw[0], (size_t)w * 3, T.color, (size_t)w * 3, w, hh, 1, 0, 0, T.st), "rgb->bgr");
                else RS_STEP(rs, gipk_cvt_bayer_to_bgr(T.raw[0], w, T.color, (size_t)w * 3, w, hh, enc - 3, 1, 0, 0, T.st), "bayer->bgr");
                color_l = T.color;
            }
        }
    }
    (void)npx;
    // rows this tile rectifies itself: its own (halo_mode 0: the halo comes from the neighbours) or all it needs (halo_mode 1)
    const int ra = rs->cfg.halo_mode ? T.e0 : T.y0, rb = rs->cfg.halo_mode ? T.e1 : T.y1;
    const size_t pitch = 2 * rs->rp;
    for (int i = 0; i < 2; i++) {
        uint8_t* dst = T.rect + (size_t)(ra - T.e0) * pitch + (size_t)i * rs->rp;
        if (T.v.distorted[i])
            RS_STEP(rs, gipk_remap_u8(mono[i], w, w, hh, dst, pitch, w, rb - ra, 1, T.v.map1[i] + (size_t)ra * w * 2, T.v.map2[i] + (size_t)ra * w, 1, 0, 0, T.st), "rectify");
        else
            RS_HIP(rs, hipMemcpy2DAsync(dst, pitch, mono[i] + (size_t)ra * w, w, w, rb - ra, hipMemcpyDeviceToDevice, T.st));
    }
    if (want_pc && color_l) {
        const int n = T.y1 - T.y0;
        if (T.v.distorted[0])
            RS_STEP(rs, gipk_remap_u8(color_l, (size_t)w * 3, w, hh, T.rect_color, (size_t)w * 3, w, n, 3, T.v.map1[0] + (size_t)T.y0 * w * 2, T.v.map2[0] + (size_t)T.y0 * w, 1, 0, 0,
                                      T.st), "rectify colour");
        else
            RS_HIP(rs, hipMemcpyAsync(T.rect_color, color_l + (size_t)T.y0 * w * 3, (size_t)n * w * 3, hipMemcpyDeviceToDevice, T.st));
    }
    return GIP_OK;
}

int tile_front(gip_rowsplit_s* rs, Tile& T)
{
    if (!T.raw[0] || !T.raw[1]) return rs_fail(rs, GIP_ERR_EMPTY_SLOT, "rowsplit: no input frame on tile " + std::to_string(T.t));
    int rc = front_body(rs, T);
    if (rc != GIP_OK) return rc;
    RS_HIP(rs, hipEventRecord(T.ev_rect, T.st));
    RS_HIP(rs, hipEventRecord(T.tev[1], T.st));
    return GIP_OK;
}

// pull the halo rows out of the neighbours' buffers: ONE copy per neighbour (both eyes)
int tile_halo(gip_rowsplit_s* rs, Tile& T)
{
    const size_t pitch = 2 * rs->rp;
    if (!rs->cfg.halo_mode) {
        if (T.t > 0 && T.y0 > T.e0) {
            Tile& U = rs->tiles[T.t - 1];
            RS_HIP(rs, hipStreamWaitEvent(T.st, U.ev_rect, 0));
            RS_HIP(rs, hipMemcpyPeerAsync(T.rect, T.device, U.rect + (size_t)(T.e0 - U.e0) * pitch, U.device, (size_t)(T.y0 - T.e0) * pitch, T.st));
        }
        if (T.t + 1 < (int)rs->tiles.size() && T.e1 > T.y1) {
            Tile& D = rs->tiles[T.t + 1];
            RS_HIP(rs, hipStreamWaitEvent(T.st, D.ev_rect, 0));
            RS_HIP(rs, hipMemcpyPeerAsync(T.rect + (size_t)(T.y1 - T.e0) * pitch, T.device, D.rect + (size_t)(T.y1 - D.e0) * pitch, D.device, (size_t)(T.e1 - T.y1) * pitch, T.st));
        }
    }
    RS_HIP(rs, hipEventRecord(T.tev[2], T.st));
    return GIP_OK;
}

int match_body(gip_rowsplit_s* rs, Tile& T)
{
    const int w = rs->cfg.width, he = T.e1 - T.e0, hown = T.y1 - T.y0;
    const size_t pitch = 2 * rs->rp;
    // prefilter of both eyes in ONE launch: the two eyes are a "batch" of two images rp bytes apart in the interleaved buffer
    uint8_t* pf[2];
    size_t pfp = 0, pfplane = 0;
    gip_stereobm_planes(T.bm_scratch, w, he, 1, &pf[0], &pf[1], &pfp, &pfplane);
    RS_STEP(rs, gipk_prefilter(T.rect, pitch, pf[0], pfp, w, he, rs->bp.prefilter_type, rs->bp.prefilter_size, rs->bp.prefilter_cap, 2, rs->rp, pfplane, T.st), "prefilter");
    RS_STEP(rs, gip_stereobm_impl(nullptr, nullptr, pfp, w, he, &rs->bp, T.d_ext, (size_t)w * 2, T.bm_scratch, T.bm_scratch_bytes, 1, pfplane, 0, T.st, nullptr), "stereobm");
    int16_t* d_own = T.d_ext + (size_t)(T.y0 - T.e0) * w;
    if (rs->spk_size > 0) {
        const int new_val = (rs->bp.min_disparity - 1) * 16;
        RS_STEP(rs, gipk_speckle_label(d_own, (size_t)w * 2, w, hown, new_val, rs->spk_diff16, T.spk_scratch, T.spk_scratch_bytes, 1, 0, T.st), "speckle label");
        if (rs->tiles.size() > 1) {
            RS_STEP(rs, gipk_speckle_border_record(d_own, (size_t)w * 2, w, hown, T.t, T.spk_scratch, T.spk_scratch_bytes, T.pub_record, T.border_roots, T.st),
                    "border record");
        }
    }
    return GIP_OK;
}

int tile_match(gip_rowsplit_s* rs, Tile& T)
{
    int rc = match_body(rs, T);
    if (rc != GIP_OK) return rc;
    if (rs->spk_size > 0 && rs->tiles.size() > 1) RS_HIP(rs, hipEventRecord(T.ev_rec, T.st));
    RS_HIP(rs, hipEventRecord(T.tev[3], T.st));
    return GIP_OK;
}

int tile_gather(gip_rowsplit_s* rs, Tile& T, bool wait_host)
{
    if (rs->spk_size > 0 && rs->tiles.size() > 1 && T.gather_kernel) {
        const void* src[16];
        for (Tile& O : rs->tiles) {
            src[O.t] = O.pub_record;
            if (O.t == T.t) continue;
            if (wait_host)
                while (rs->rec_gen[O.t].load(std::memory_order_acquire) < rs->frame) std::this_thread::yield();
            RS_HIP(rs, hipStreamWaitEvent(T.st, O.ev_rec, 0));
        }
        RS_STEP(rs, gipk_speckle_border_gather(src, (int)rs->tiles.size(), rs->cfg.width, T.records, T.st), "border gather");
    } else if (rs->spk_size > 0 && rs->tiles.size() > 1) {
        for (Tile& O : rs->tiles) {
            uint8_t* dst = T.records + (size_t)O.t * rs->rec_bytes;
            if (O.t == T.t) {
                RS_HIP(rs, hipMemcpyAsync(dst, T.pub_record, (size_t)20 * rs->cfg.width, hipMemcpyDeviceToDevice, T.st));
                continue;
            }
            if (wait_host)
                while (rs->rec_gen[O.t].load(std::memory_order_acquire) < rs->frame) std::this_thread::yield();
            RS_HIP(rs, hipStreamWaitEvent(T.st, O.ev_rec, 0));
            RS_HIP(rs, hipMemcpyPeerAsync(dst, T.device, O.pub_record, O.device, (size_t)20 * rs->cfg.width, T.st));
        }
    }
    RS_HIP(rs, hipEventRecord(T.tev[4], T.st));
    return GIP_OK;
}

int post_body(gip_rowsplit_s* rs, Tile& T)
{
    const int w = rs->cfg.width, hown = T.y1 - T.y0;
    int16_t* d_own = T.d_ext + (size_t)(T.y0 - T.e0) * w;
    if (rs->spk_size > 0) {
        const int new_val = (rs->bp.min_disparity - 1) * 16;
        if (rs->tiles.size() > 1)
            RS_STEP(rs, gipk_speckle_border_merge(T.records, (int)rs->tiles.size(), T.t, w, hown, new_val, rs->spk_diff16, T.spk_scratch, T.spk_scratch_bytes, T.border_roots, T.st),
                    "border merge");
        RS_STEP(rs, gipk_speckle_apply(d_own, (size_t)w * 2, w, hown, new_val, rs->spk_size, T.spk_scratch, T.spk_scratch_bytes, 1, 0, T.st), "speckle apply");
    }
    if (T.f32 || T.pc) {
        const uint8_t* col = nullptr;
        size_t cpitch = 0;
        int cn = 3;
        if (T.pc) {
            if (T.rect_color) { col = T.rect_color; cpitch = (size_t)w * 3; }
            else { col = T.rect + (size_t)(T.y0 - T.e0) * 2 * rs->rp; cpitch = 2 * rs->rp; cn = 1; }   // mono input: the rectified left rows, replicated (GRAY2BGR)
        }
        RS_STEP(rs, gip_disparity_outputs_impl(d_own, (size_t)w * 2, col, cpitch, cn, T.f32, T.pc, w, hown, T.y0, T.v.Q, T.v.cx_l, T.v.cx_r, rs->bp.min_disparity, 1, 0, 0, T.st,
                                               T.w_tab, rs->bp.num_disparities, true), "disparity outputs");
    }
    return GIP_OK;
}

int tile_post(gip_rowsplit_s* rs, Tile& T)
{
    int rc = post_body(rs, T);
    if (rc != GIP_OK) return rc;
    RS_HIP(rs, hipEventRecord(T.tev[5], T.st));
    RS_HIP(rs, hipEventRecord(T.ev_done, T.st));
    T.ran = true;
    return GIP_OK;
}

// the whole frame of one tile, as its enqueueing thread runs it
int tile_frame_threaded(gip_rowsplit_s* rs, Tile& T)
{
    int rc;
    if ((rc = tile_begin(rs, T)) != GIP_OK) return rc;
    rc = tile_front(rs, T);
    rs->rect_gen[T.t].store(rs->frame, std::memory_order_release);      // (also on failure: nobody may wait for ever)
    if (rc != GIP_OK) { rs->rec_gen[T.t].store(rs->frame, std::memory_order_release); return rc; }
    // the neighbours' ev_rect must have been RECORDED for this frame before this thread waits on them
    if (!rs->cfg.halo_mode) {
        if (T.t > 0)
            while (rs->rect_gen[T.t - 1].load(std::memory_order_acquire) < rs->frame) std::this_thread::yield();
        if (T.t + 1 < (int)rs->tiles.size())
            while (rs->rect_gen[T.t + 1].load(std::memory_order_acquire) < rs->frame) std::this_thread::yield();
    }
    rc = tile_halo(rs, T);
    if (rc == GIP_OK) rc = tile_match(rs, T);
    rs->rec_gen[T.t].store(rs->frame, std::memory_order_release);
    if (rc != GIP_OK) return rc;
    if ((rc = tile_gather(rs, T, true)) != GIP_OK) return rc;
    return tile_post(rs, T);
}

void worker_main(gip_rowsplit_s* rs, int t)
{
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(rs->mu);
            rs->cv_go.wait(lk, [&] { return rs->quit || rs->gen != seen; });
            if (rs->quit) return;
            seen = rs->gen;
        }
        const int rc = tile_frame_threaded(rs, rs->tiles[t]);
        {
            std::lock_guard<std::mutex> lk(rs->mu);
            rs->worker_rc[t] = rc;
            if (--rs->pending == 0) rs->cv_done.notify_all();
        }
    }
}

void free_tile(Tile& T)
{
    (void)hipSetDevice(T.device);
    if (T.st) (void)hipStreamSynchronize(T.st);
    void* bufs[] = {T.own_raw[0], T.own_raw[1], T.mono[0], T.mono[1], T.color, T.rect, T.rect_color, T.bm_scratch, T.d_ext, T.spk_scratch, T.records, T.pub_record, T.border_roots, T.w_tab,
                    T.f32, T.pc};
    for (void* p : bufs)
        if (p) (void)hipFree(p);
    for (int i = 0; i < 2; i++)
        if (T.stage[i]) (void)hipHostFree(T.stage[i]);
    hipEvent_t evs[] = {T.sev, T.ev_rect, T.ev_rec, T.ev_done, T.tev[0], T.tev[1], T.tev[2], T.tev[3], T.tev[4], T.tev[5]};
    for (hipEvent_t e : evs)
        if (e) (void)hipEventDestroy(e);
    if (T.st) (void)hipStreamDestroy(T.st);
    if (T.h) (void)gip_destroy(T.h);
}

void destroy(gip_rowsplit_s* rs)
{
    if (!rs->workers.empty()) {
        {
            std::lock_guard<std::mutex> lk(rs->mu);
            rs->quit = true;
        }
        rs->cv_go.notify_all();
        for (std::thread& th : rs->workers) th.join();
    }
    for (Tile& T : rs->tiles) free_tile(T);
    delete[] rs->rect_gen;
    delete[] rs->rec_gen;
    delete rs;
}

int timing_of(gip_rowsplit_s* rs, Tile& T, gip_rowsplit_timing* out)
{
    if (!T.ran) return rs_fail(rs, GIP_ERR_INVALID_ARG, "rowsplit: no frame has run");
    float* dst[5] = {&out->front_ms, &out->halo_ms, &out->match_ms, &out->gather_ms, &out->post_ms};
    for (int i = 0; i < 5; i++) RS_HIP(rs, hipEventElapsedTime(dst[i], T.tev[i], T.tev[i + 1]));
    RS_HIP(rs, hipEventElapsedTime(&out->total_ms, T.tev[0], T.tev[5]));
    return GIP_OK;
}

}  // namespace

extern "C" {

// the plan alone (host only, no device): rows of tile `tile` of an n_tiles split of a frame `height` rows tall for these matcher parameters
int gip_rowsplit_plan_rows(int height, int n_tiles, const gip_bm_params* params, int tile, int* y0, int* y1, int* e0, int* e1, int* halo)
{
    if (!params || height <= 0 || n_tiles < 1 || n_tiles > 16 || tile < 0 || tile >= n_tiles) return GIP_ERR_INVALID_ARG;
    const int hl = halo_rows(*params), base = height / n_tiles, rem = height % n_tiles;
    if (n_tiles > 1 && base < hl + 2) return GIP_ERR_INVALID_ARG;        // a tile must be taller than the rows it lends to its neighbours
    const int a = tile * base + (tile < rem ? tile : rem), b = a + base + (tile < rem ? 1 : 0);
    int x0, x1;
    extended_range(height, a, b, hl, &x0, &x1);
    if (y0) *y0 = a;
    if (y1) *y1 = b;
    if (e0) *e0 = x0;
    if (e1) *e1 = x1;
    if (halo) *halo = hl;
    return GIP_OK;
}

int gip_rowsplit_open(const gip_camera_info* left, const gip_camera_info* right, const gip_bm_params* params, int max_speckle_size, double max_speckle_diff,
                      const gip_rowsplit_config* cfg, gip_rowsplit* out)
{
    if (!left || !right || !params || !cfg || !out) return GIP_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->encoding < 0 || cfg->encoding > 6 || cfg->n_tiles < 1 || cfg->n_tiles > 16 || max_speckle_size < 0 ||
        max_speckle_diff < 0.0 || (cfg->halo_mode != 0 && cfg->halo_mode != 1) || cfg->gather_mode < 0 || cfg->gather_mode > 2)
        return GIP_ERR_INVALID_ARG;
    const int ndev = gip_device_count();
    if (ndev <= 0) return GIP_ERR_NO_DEVICE;
    gip_rowsplit_s* rs = new (std::nothrow) gip_rowsplit_s();
    if (!rs) return GIP_ERR_HIP;
    rs->cfg = *cfg;
    rs->bp = *params;
    rs->bp.speckle_window_size = 0;     // the node-level filter (gip_filter_speckles semantics) is the one that runs, as in gip_batch_run
    rs->spk_size = max_speckle_size;
    rs->spk_diff16 = (int)(max_speckle_diff * 16.0);
    rs->halo = halo_rows(rs->bp);
    const int w = cfg->width, hh = cfg->height, N = cfg->n_tiles;
    rs->rp = gip_align_up((size_t)w, 64);
    rs->rec_bytes = gipk_speckle_border_record_bytes(w);
    auto fail = [&](int code) { destroy(rs); return code; };
    if (gip_validate_bm_params(&rs->bp, w, hh) != GIP_OK) return fail(GIP_ERR_INVALID_ARG);
    // the plan: contiguous, near-equal row ranges; a tile must be at least as tall as the halo it lends to its neighbours
    rs->tiles.resize(N);
    for (int t = 0; t < N; t++) {
        Tile& T = rs->tiles[t];
        T.t = t;
        if (gip_rowsplit_plan_rows(hh, N, &rs->bp, t, &T.y0, &T.y1, &T.e0, &T.e1, nullptr) != GIP_OK) return fail(GIP_ERR_INVALID_ARG);
        T.device = cfg->devices[t] >= 0 ? cfg->devices[t] : t % ndev;
        if (T.device >= ndev) return fail(GIP_ERR_INVALID_ARG);
    }
    const bool want_f32 = (cfg->outputs & GIP_TOPIC_DISPARITY) != 0, want_pc = (cfg->outputs & GIP_TOPIC_POINTCLOUD) != 0;
    const int cn = raw_channels(cfg->encoding);
    for (Tile& T : rs->tiles) {
        int rc = gip_create(T.device, &T.h);
        if (rc != GIP_OK) return fail(rc);
        if ((rc = gip_init_stereo_model(T.h, left, right)) != GIP_OK || (rc = gip_set_params(T.h, params)) != GIP_OK) return fail(rc);
        if ((rc = gip_handle_view(T.h, &T.v)) != GIP_OK) return fail(rc);
        if (T.v.width != w || T.v.height != hh) return fail(GIP_ERR_INVALID_ARG);
        if (hipSetDevice(T.device) != hipSuccess) return fail(GIP_ERR_HIP);
        bool ok = hipStreamCreateWithFlags(&T.st, hipStreamNonBlocking) == hipSuccess;
        const int he = T.e1 - T.e0, hown = T.y1 - T.y0;
        const size_t npx = (size_t)w * hh;
        if (cfg->encoding != 0) {
            ok = ok && hipMalloc((void**)&T.mono[0], npx) == hipSuccess && hipMalloc((void**)&T.mono[1], npx) == hipSuccess;
            if (want_pc && cfg->encoding != 1) ok = ok && hipMalloc((void**)&T.color, npx * 3) == hipSuccess;
            if (want_pc) ok = ok && hipMalloc((void**)&T.rect_color, (size_t)hown * w * 3) == hipSuccess;
        }
        ok = ok && hipMalloc((void**)&T.rect, (size_t)he * 2 * rs->rp) == hipSuccess;
        T.bm_scratch_bytes = gipk_stereobm_scratch_bytes(w, he, 1, &rs->bp);
        ok = ok && hipMalloc(&T.bm_scratch, T.bm_scratch_bytes) == hipSuccess;
        ok = ok && hipMalloc((void**)&T.d_ext, (size_t)he * w * 2) == hipSuccess;
        if (rs->spk_size > 0) {
            T.spk_scratch_bytes = gipk_speckle_scratch_bytes(w, hown, 1);
            ok = ok && hipMalloc(&T.spk_scratch, T.spk_scratch_bytes) == hipSuccess;
            if (N > 1) {
                ok = ok && hipMalloc((void**)&T.records, rs->rec_bytes * N) == hipSuccess && hipMalloc((void**)&T.pub_record, rs->rec_bytes) == hipSuccess &&
                     hipMalloc((void**)&T.border_roots, (size_t)2 * w * 4) == hipSuccess;
            }
        }
        if (want_f32 || want_pc) ok = ok && hipMalloc((void**)&T.f32, (size_t)hown * w * 4) == hipSuccess;
        if (want_pc) ok = ok && hipMalloc((void**)&T.pc, (size_t)hown * w * 32) == hipSuccess;
        if (want_pc) ok = ok && hipMalloc(&T.w_tab, gip_w_table_bytes(rs->bp.num_disparities)) == hipSuccess &&
                          gip_w_table_build(T.w_tab, T.v.Q, T.v.cx_l, T.v.cx_r, rs->bp.min_disparity, rs->bp.num_disparities, T.st) == GIP_OK;
        ok = ok && hipEventCreateWithFlags(&T.ev_rect, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&T.ev_rec, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&T.ev_done, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 6; i++) ok = ok && hipEventCreate(&T.tev[i]) == hipSuccess;
        ok = ok && hipEventCreate(&T.sev) == hipSuccess;
        (void)cn;
        if (!ok) return fail(GIP_ERR_HIP);
    }
    // peer access between the devices of neighbouring tiles and of all tiles (records); failing to enable it is not fatal:
    // hipMemcpyPeerAsync then stages through the host
    for (Tile& A : rs->tiles)
        for (Tile& B : rs->tiles)
            if (A.device != B.device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, A.device, B.device) == hipSuccess && can) {
                    (void)hipSetDevice(A.device);
                    hipError_t e = hipDeviceEnablePeerAccess(B.device, 0);
                    if (e != hipSuccess) (void)hipGetLastError();      // already enabled, or refused: both fine
                }
            }
    for (Tile& A : rs->tiles) {
        bool same = true;
        for (Tile& B : rs->tiles) same = same && A.device == B.device;
        A.gather_kernel = cfg->gather_mode == 2 || (cfg->gather_mode == 0 && same);
    }
    rs->rect_gen = new (std::nothrow) std::atomic<uint64_t>[N];
    rs->rec_gen = new (std::nothrow) std::atomic<uint64_t>[N];
    if (!rs->rect_gen || !rs->rec_gen) return fail(GIP_ERR_HIP);
    for (int t = 0; t < N; t++) { rs->rect_gen[t].store(0); rs->rec_gen[t].store(0); }
    rs->worker_rc.assign(N, GIP_OK);
    if (N > 1 && cfg->host_threads != 1)
        for (int t = 0; t < N; t++) rs->workers.emplace_back(worker_main, rs, t);
    *out = rs;
    return GIP_OK;
}

int gip_rowsplit_close(gip_rowsplit rs)
{
    if (!rs) return GIP_ERR_INVALID_ARG;
    destroy(rs);
    return GIP_OK;
}

const char* gip_rowsplit_last_error(gip_rowsplit rs)
{
    if (!rs) return "null rowsplit handle";
    std::lock_guard<std::mutex> lk(rs->err_mu);
    return rs->err.c_str();
}

int gip_rowsplit_plan(gip_rowsplit rs, int tile, int* y0, int* y1, int* e0, int* e1, int* device)
{
    if (!rs || tile < 0 || tile >= (int)rs->tiles.size()) return GIP_ERR_INVALID_ARG;
    const Tile& T = rs->tiles[tile];
    if (y0) *y0 = T.y0;
    if (y1) *y1 = T.y1;
    if (e0) *e0 = T.e0;
    if (e1) *e1 = T.e1;
    if (device) *device = T.device;
    return GIP_OK;
}

int gip_rowsplit_set_inputs_device(gip_rowsplit rs, int tile, const uint8_t* left_raw, const uint8_t* right_raw)
{
    if (!rs || tile < 0 || tile >= (int)rs->tiles.size() || !left_raw || !right_raw) return GIP_ERR_INVALID_ARG;
    rs->tiles[tile].raw[0] = left_raw;
    rs->tiles[tile].raw[1] = right_raw;
    return GIP_OK;
}

int gip_rowsplit_upload(gip_rowsplit rs, const void* left_raw, const void* right_raw, size_t step)
{
    if (!rs || !left_raw || !right_raw) return GIP_ERR_INVALID_ARG;
    const int w = rs->cfg.width, hh = rs->cfg.height, cn = raw_channels(rs->cfg.encoding);
    const size_t row = (size_t)w * cn, bytes = row * hh;
    if (step == 0) step = row;
    if (step < row) return rs_fail(rs, GIP_ERR_INVALID_ARG, "rowsplit upload: step smaller than a row");
    // one pinned copy of the frame (tile 0 owns it), then one H2D copy per tile on the tile's own stream: on N devices the N copies
    // run in parallel on N host links
    Tile& P = rs->tiles[0];
    const void* src[2] = {left_raw, right_raw};
    for (int i = 0; i < 2; i++) {
        if (!P.stage[i]) RS_HIP(rs, hipHostMalloc((void**)&P.stage[i], bytes, hipHostMallocDefault));
    }
    // the staging buffers may still be read by the previous frame's copies
    for (Tile& T : rs->tiles) {
        RS_HIP(rs, hipSetDevice(T.device));
        RS_HIP(rs, hipStreamSynchronize(T.st));
    }
    for (int i = 0; i < 2; i++) {
        if (step == row) std::memcpy(P.stage[i], src[i], bytes);
        else
            for (int y = 0; y < hh; y++) std::memcpy(P.stage[i] + (size_t)y * row, (const uint8_t*)src[i] + (size_t)y * step, row);
    }
    for (Tile& T : rs->tiles) {
        RS_HIP(rs, hipSetDevice(T.device));
        for (int i = 0; i < 2; i++) {
            if (!T.own_raw[i]) RS_HIP(rs, hipMalloc((void**)&T.own_raw[i], bytes));
            RS_HIP(rs, hipMemcpyAsync(T.own_raw[i], P.stage[i], bytes, hipMemcpyHostToDevice, T.st));
            T.raw[i] = T.own_raw[i];
        }
    }
    return GIP_OK;
}

int gip_rowsplit_run(gip_rowsplit rs)
{
    if (!rs) return GIP_ERR_INVALID_ARG;
    rs->frame++;
    int rc = GIP_OK;
    if (!rs->workers.empty()) {
        {
            std::lock_guard<std::mutex> lk(rs->mu);
            rs->pending = (int)rs->tiles.size();
            rs->gen++;
        }
        rs->cv_go.notify_all();
        std::unique_lock<std::mutex> lk(rs->mu);
        rs->cv_done.wait(lk, [&] { return rs->pending == 0; });
        for (int r : rs->worker_rc)
            if (r != GIP_OK && rc == GIP_OK) rc = r;
    } else {
        // one enqueueing thread: phase by phase over the tiles, so that every event is recorded before anybody waits on it
        for (Tile& T : rs->tiles)
            if (rc == GIP_OK) rc = tile_begin(rs, T);
        for (Tile& T : rs->tiles)
            if (rc == GIP_OK) rc = tile_front(rs, T);
        for (Tile& T : rs->tiles)
            if (rc == GIP_OK) rc = tile_halo(rs, T);
        for (Tile& T : rs->tiles)
            if (rc == GIP_OK) rc = tile_match(rs, T);
        for (Tile& T : rs->tiles)
            if (rc == GIP_OK) rc = tile_gather(rs, T, false);
        for (Tile& T : rs->tiles)
            if (rc == GIP_OK) rc = tile_post(rs, T);
    }
    if (rc == GIP_OK) rs->frames_done = true;
    else {
        // a failed enqueue leaves events unrecorded: drain everything so that the next frame starts from a clean state
        for (Tile& T : rs->tiles) {
            (void)hipSetDevice(T.device);
            (void)hipStreamSynchronize(T.st);
        }
        rs->frames_done = false;
    }
    return rc;
}

int gip_rowsplit_wait(gip_rowsplit rs)
{
    if (!rs) return GIP_ERR_INVALID_ARG;
    for (Tile& T : rs->tiles) {
        RS_HIP(rs, hipSetDevice(T.device));
        RS_HIP(rs, hipStreamSynchronize(T.st));
    }
    return GIP_OK;
}

int gip_rowsplit_run_serial(gip_rowsplit rs, gip_rowsplit_timing* timing)
{
    if (!rs) return GIP_ERR_INVALID_ARG;
    int rc = gip_rowsplit_wait(rs);
    if (rc != GIP_OK) return rc;
    rs->frame++;
    rs->frames_done = false;     // everything is drained: no cross-frame waits needed
    std::vector<gip_rowsplit_timing> tm(rs->tiles.size(), gip_rowsplit_timing{});
    // one phase of one tile with the device(s) otherwise idle: start event, the phase (which records its own end event tev[end]), drain
    auto timed = [&](Tile& T, int (*phase)(gip_rowsplit_s*, Tile&), int end, float gip_rowsplit_timing::*field) -> int {
        RS_HIP(rs, hipSetDevice(T.device));
        RS_HIP(rs, hipEventRecord(T.sev, T.st));
        int r = phase(rs, T);
        if (r != GIP_OK) return r;
        RS_HIP(rs, hipStreamSynchronize(T.st));
        float ms = 0.f;
        RS_HIP(rs, hipEventElapsedTime(&ms, T.sev, T.tev[end]));
        tm[T.t].*field = ms;
        return GIP_OK;
    };
    // the tile that goes first after a drain finds the device idle (clocks down, caches cold): rotate the order from call to call so that
    // the effect does not always land on tile 0 (callers take the median over several calls)
    const int N = (int)rs->tiles.size(), first = (int)(rs->frame % (uint64_t)N);
    auto at = [&](int i) -> Tile& { return rs->tiles[(first + i) % N]; };
    for (Tile& T : rs->tiles)
        if ((rc = tile_begin(rs, T)) != GIP_OK) return rc;
    for (int i = 0; i < N; i++)
        if ((rc = timed(at(i), tile_front, 1, &gip_rowsplit_timing::front_ms)) != GIP_OK) return rc;
    for (int i = 0; i < N; i++)
        if ((rc = timed(at(i), tile_halo, 2, &gip_rowsplit_timing::halo_ms)) != GIP_OK) return rc;
    for (int i = 0; i < N; i++)
        if ((rc = timed(at(i), tile_match, 3, &gip_rowsplit_timing::match_ms)) != GIP_OK) return rc;
    for (int i = 0; i < N; i++)
        if ((rc = timed(at(i), [](gip_rowsplit_s* r, Tile& X) { return tile_gather(r, X, false); }, 4, &gip_rowsplit_timing::gather_ms)) != GIP_OK) return rc;
    for (int i = 0; i < N; i++)
        if ((rc = timed(at(i), tile_post, 5, &gip_rowsplit_timing::post_ms)) != GIP_OK) return rc;
    rs->frames_done = true;
    for (Tile& T : rs->tiles) {
        gip_rowsplit_timing& x = tm[T.t];
        x.total_ms = x.front_ms + x.halo_ms + x.match_ms + x.gather_ms + x.post_ms;
        T.serial_tm = x;
        T.serial_valid = true;
        if (timing) timing[T.t] = x;
    }
    return GIP_OK;
}

int gip_rowsplit_last_timing(gip_rowsplit rs, int tile, gip_rowsplit_timing* out)
{
    if (!rs || !out || tile < 0 || tile >= (int)rs->tiles.size()) return GIP_ERR_INVALID_ARG;
    return timing_of(rs, rs->tiles[tile], out);
}

int gip_rowsplit_outputs_device(gip_rowsplit rs, int tile, const int16_t** disparity, const float** disparity_f32, const uint8_t** pointcloud)
{
    if (!rs || tile < 0 || tile >= (int)rs->tiles.size()) return GIP_ERR_INVALID_ARG;
    const Tile& T = rs->tiles[tile];
    if (disparity) *disparity = T.d_ext + (size_t)(T.y0 - T.e0) * rs->cfg.width;
    if (disparity_f32) *disparity_f32 = (rs->cfg.outputs & GIP_TOPIC_DISPARITY) ? T.f32 : nullptr;
    if (pointcloud) *pointcloud = T.pc;
    return GIP_OK;
}

int gip_rowsplit_download(gip_rowsplit rs, int16_t* disparity, float* disparity_f32, uint8_t* pointcloud)
{
    if (!rs) return GIP_ERR_INVALID_ARG;
    const size_t w = (size_t)rs->cfg.width;
    if (disparity_f32 && !(rs->cfg.outputs & GIP_TOPIC_DISPARITY)) return rs_fail(rs, GIP_ERR_INVALID_ARG, "rowsplit download: float disparity was not requested at open");
    if (pointcloud && !(rs->cfg.outputs & GIP_TOPIC_POINTCLOUD)) return rs_fail(rs, GIP_ERR_INVALID_ARG, "rowsplit download: the cloud was not requested at open");
    // every tile copies its rows straight into the caller's frame ("D2H directly from each GPU into one host image")
    for (Tile& T : rs->tiles) {
        RS_HIP(rs, hipSetDevice(T.device));
        const size_t n = (size_t)(T.y1 - T.y0) * w;
        if (disparity) RS_HIP(rs, hipMemcpyAsync(disparity + (size_t)T.y0 * w, T.d_ext + (size_t)(T.y0 - T.e0) * w, n * 2, hipMemcpyDeviceToHost, T.st));
        if (disparity_f32) RS_HIP(rs, hipMemcpyAsync(disparity_f32 + (size_t)T.y0 * w, T.f32, n * 4, hipMemcpyDeviceToHost, T.st));
        if (pointcloud) RS_HIP(rs, hipMemcpyAsync(pointcloud + (size_t)T.y0 * w * 32, T.pc, n * 32, hipMemcpyDeviceToHost, T.st));
    }
    return gip_rowsplit_wait(rs);
}

}  // extern "C"
