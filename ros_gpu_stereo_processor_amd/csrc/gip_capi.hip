// gip_capi.hip -- the handle API of libgpuimageproc_hip.so: a MI355X-native stand-in for the reference's
// gpuimageproc::GpuStereoProcessor (reference include/gpuimageproc/GPUStereoProcessor.h:63-126,
// src/GPUStereoProcessor.cpp) behind a C ABI.
//
// What is deliberately different from the reference:
//   * buffers are device-resident HBM allocations fed by pinned hipMemcpyAsync, not mapped host memory
//     (cv::cuda::HostMem::SHARED, GPUStereoProcessor.cpp:174-188) -- on a discrete GPU the latter makes every
//     kernel PCIe-bound;
//   * every slot carries a "ready" event, so a consumer on the other side's stream waits for its producer (the
//     reference has an unsynchronised cross-stream read between right-rectify and disparity, :248 / :283);
//   * the disparity is the CPU matcher's int16 x16 map plus a real float image; the speckle filter runs on the
//     device instead of draining the stream for a CPU flood fill (:356-365);
//   * errors are status codes, never exceptions/asserts.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "gip_internal.h"
#include "gip_model.h"
#include "bm_common.h"
#include "gip_handle_view.h"

// ------------------------------------------------------------------------------------------------ error text
static thread_local char g_hip_err[512] = "";

void gip_note_hip_error(hipError_t e, const char* what, const char* file, int line)
{
    std::snprintf(g_hip_err, sizeof g_hip_err, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
}
const char* gip_last_hip_error_text() { return g_hip_err; }

namespace {

inline int elem_size(int type)
{
    int depth = type & 7, cn = (type >> 3) + 1;
    int ds = (depth == GIP_8U) ? 1 : (depth == GIP_16U || depth == GIP_16S) ? 2 : (depth == GIP_32F) ? 4 : 0;
    return ds * cn;
}

struct Slot {
    void* d = nullptr;
    size_t pitch = 0, cap = 0;
    int rows = 0, cols = 0, type = -1;
    hipEvent_t ready = nullptr;   // recorded after the last write
    hipEvent_t read = nullptr;    // recorded after the last read from a foreign stream
    hipStream_t writer = nullptr;
    bool has_reader = false;
    // pinned staging for uploadMat: two buffers used alternately, so that staging frame i+1 (a host memcpy) does not wait for the
    // H2D copy of frame i
    void* stage[2] = {nullptr, nullptr};
    size_t stage_cap[2] = {0, 0};
    hipEvent_t stage_done[2] = {nullptr, nullptr};
    bool stage_busy[2] = {false, false};
    int stage_next = 0;
};

enum SenderKind { SEND_IMAGE, SEND_DISPARITY, SEND_POINTS };

struct Sender {
    SenderKind kind;
    std::atomic<bool> sent{false};
    void* host = nullptr;  // pinned
    size_t host_bytes = 0;
    gip_image_msg img{};
    gip_disparity_msg dmsg{};
    gip_pointcloud2_msg pc{};
    gip_image_cb icb = nullptr;
    gip_disparity_cb dcb = nullptr;
    gip_pointcloud2_cb pcb = nullptr;
    void* user = nullptr;
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

}  // namespace

struct gip_processor {
    int device = 0;
    hipStream_t l_strm = nullptr, r_strm = nullptr;
    hipEvent_t match_fork = nullptr, match_join = nullptr;   // gip_batch_desc::matcher_stream fences
    bool front_pending = false;                              // gip_batch_run: a FRONT phase waits for its BACK phase
    hipEvent_t front_join = nullptr;                         //   ... the matcher's completion event of that FRONT phase (one of the events above / tev)
    int front_cost_u16 = 1;                                  //   ... and the type of its min-SAD plane
    hipEvent_t back_done = nullptr;                          // recorded at the end of a BACK phase: the next FRONT phase (possibly on another stream) waits for it
    bool back_done_valid = false;
    std::unordered_map<int, Slot> slots;
    GipStereoModel model;
    gip_bm_params params{};
    int max_speckle_size = 0;
    double max_speckle_diff = 0.0;
    int refine_disparity = 0;
    std::string l_enc, r_enc, l_cam_name, r_cam_name;
    std::string err;
    // device copies of the rectification tables
    int16_t* d_map1[2] = {nullptr, nullptr};
    uint16_t* d_map2[2] = {nullptr, nullptr};
    int map_w = 0, map_h = 0;
    bool maps_owned = true;
    DevBuf bm_scratch, spk_scratch, pc_dev;
    std::vector<std::unique_ptr<Sender>> senders;
    std::vector<std::pair<void*, size_t>> pinned_pool;
    // batch pipeline
    DevBuf b_mono[2], b_color, b_rect[2], b_rect_color, b_f32;
    // table of everything the cloud kernel derives from the disparity value alone (k_post.hip: gip_w_table_build), rebuilt when its key changes
    DevBuf w_tab;
    double w_key[6] = {0, 0, 0, 0, 0, 0};      // Q23, Q32, Q33, cx_l - cx_r, min_disparity, num_disparities
    bool w_valid = false;
    // per-run timing events of the batch pipeline: 5 per run (bm0, bm1, total0, total1, end of rectify+prefilter+match+validate), grown on demand
    std::vector<hipEvent_t> tev;
    int timing_runs = 0;
};

namespace {

#define H_CHECK(h)                                  \
    do {                                            \
        if (!(h)) return GIP_ERR_INVALID_ARG;       \
    } while (0)

#define H_HIP(h, expr)                                                             \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            gip_note_hip_error(_e, #expr, __FILE__, __LINE__);                     \
            (h)->err = gip_last_hip_error_text();                                  \
            return GIP_ERR_HIP;                                                    \
        }                                                                          \
    } while (0)

int fail(gip_handle h, int code, const std::string& msg)
{
    h->err = msg;
    return code;
}

int propagate(gip_handle h, int rc, const char* what)
{
    if (rc == GIP_OK) return rc;
    if (rc == GIP_ERR_HIP) h->err = gip_last_hip_error_text();
    else h->err = std::string(what) + ": " + gip_status_string(rc);
    return rc;
}

// The two side streams (GpuStereoProcessor::getStream, GPUStereoProcessor.cpp:190-200) are created on FIRST USE, and as a pair that was measured
// to run side by side (gip_concurrent_streams_create): the runtime multiplexes all streams of a process onto a handful of hardware queues, and
// streams that share a queue serialise.  A handle that only ever runs gip_batch_run on the caller's streams (throughput pipelines, the streamer's
// lanes, row-split tiles) therefore never parks two idle streams on those queues.  If the pair cannot be created the null stream serves
// (correct, just not concurrent).
hipStream_t stream_of(gip_handle h, int slot_or_side)
{
    if (!h->l_strm && !h->r_strm) {
        gip_stream pair[2] = {nullptr, nullptr};
        if (gip_concurrent_streams_create(h->device, 2, pair, nullptr) == GIP_OK) {
            h->l_strm = (hipStream_t)pair[0];
            h->r_strm = (hipStream_t)pair[1];
        } else {
            (void)hipGetLastError();
        }
    }
    return (slot_or_side & GIP_SIDE_L) ? h->l_strm : h->r_strm;
}

int ensure_dev(gip_handle h, DevBuf& b, size_t bytes)
{
    if (b.cap >= bytes) return GIP_OK;
    if (b.p) {
        // the buffer may still be in use by queued work (a clone has no side streams: hipFree itself waits for the device)
        if (h->l_strm) H_HIP(h, hipStreamSynchronize(h->l_strm));
        if (h->r_strm) H_HIP(h, hipStreamSynchronize(h->r_strm));
        H_HIP(h, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    H_HIP(h, hipMalloc(&b.p, bytes));
    b.cap = bytes;
    return GIP_OK;
}

// (re)shape a slot; keeps the allocation when large enough
int slot_create(gip_handle h, int id, int rows, int cols, int type, Slot** out)
{
    Slot& s = h->slots[id];
    const size_t row_bytes = (size_t)cols * elem_size(type);
    // tight rows: 1-D hipMemcpyAsync moves a whole image in one DMA (a pitched 2-D copy of a 1242x375 image costs 2.7 ms on
    // this runtime, the 1-D copy 0.03 ms); the kernels take any pitch
    const size_t pitch = row_bytes ? row_bytes : 1;
    const size_t need = pitch * (size_t)(rows > 0 ? rows : 1);
    if (!s.ready) {
        H_HIP(h, hipEventCreateWithFlags(&s.ready, hipEventDisableTiming));
        H_HIP(h, hipEventCreateWithFlags(&s.read, hipEventDisableTiming));
        H_HIP(h, hipEventCreateWithFlags(&s.stage_done[0], hipEventDisableTiming));
        H_HIP(h, hipEventCreateWithFlags(&s.stage_done[1], hipEventDisableTiming));
    }
    if (s.cap < need) {
        if (s.d) {
            if (h->l_strm) H_HIP(h, hipStreamSynchronize(h->l_strm));
            if (h->r_strm) H_HIP(h, hipStreamSynchronize(h->r_strm));
            H_HIP(h, hipFree(s.d));
            s.d = nullptr;
        }
        H_HIP(h, hipMalloc(&s.d, need));
        s.cap = need;
    }
    s.pitch = pitch;
    s.rows = rows;
    s.cols = cols;
    s.type = type;
    *out = &s;
    return GIP_OK;
}

Slot* slot_get(gip_handle h, int id)
{
    auto it = h->slots.find(id);
    if (it == h->slots.end() || !it->second.d || it->second.type < 0) return nullptr;
    return &it->second;
}

// stream-ordering helpers: a reader on a foreign stream waits for the writer; a writer waits for foreign readers
int begin_read(gip_handle h, Slot* s, hipStream_t st)
{
    if (s->writer && s->writer != st) H_HIP(h, hipStreamWaitEvent(st, s->ready, 0));
    return GIP_OK;
}
int end_read(gip_handle h, Slot* s, hipStream_t st)
{
    if (s->writer && s->writer != st) {
        H_HIP(h, hipEventRecord(s->read, st));
        s->has_reader = true;
    }
    return GIP_OK;
}
int begin_write(gip_handle h, Slot* s, hipStream_t st)
{
    if (s->has_reader) {
        H_HIP(h, hipStreamWaitEvent(st, s->read, 0));
        s->has_reader = false;
    }
    if (s->writer && s->writer != st) H_HIP(h, hipStreamWaitEvent(st, s->ready, 0));
    return GIP_OK;
}
int end_write(gip_handle h, Slot* s, hipStream_t st)
{
    H_HIP(h, hipEventRecord(s->ready, st));
    s->writer = st;
    return GIP_OK;
}

void* pinned_acquire(gip_handle h, size_t bytes)
{
    for (size_t i = 0; i < h->pinned_pool.size(); i++)
        if (h->pinned_pool[i].second >= bytes) {
            void* p = h->pinned_pool[i].first;
            h->pinned_pool.erase(h->pinned_pool.begin() + i);
            return p;
        }
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

int bayer_pattern(const std::string& enc)
{
    if (enc == "bayer_rggb8") return GIP_BAYER_RGGB;
    if (enc == "bayer_bggr8") return GIP_BAYER_BGGR;
    if (enc == "bayer_gbrg8") return GIP_BAYER_GBRG;
    if (enc == "bayer_grbg8") return GIP_BAYER_GRBG;
    return -1;
}

int encoding_type(const std::string& enc)
{
    if (enc == "mono8" || bayer_pattern(enc) >= 0) return GIP_8UC1;
    if (enc == "bgr8" || enc == "rgb8") return GIP_8UC3;
    if (enc == "bgra8" || enc == "rgba8") return GIP_8UC4;
    if (enc == "mono16") return GIP_16UC1;
    if (enc == "32FC1") return GIP_32FC1;
    return -1;
}

void copy_cstr(char* dst, size_t n, const char* src)
{
    std::snprintf(dst, n, "%s", src ? src : "");
}

int upload_maps(gip_handle h)
{
    const int w = h->model.left.info.width, hh = h->model.left.info.height;
    const size_t n = (size_t)w * hh;
    if (!h->maps_owned) {
        for (int i = 0; i < 2; i++) { h->d_map1[i] = nullptr; h->d_map2[i] = nullptr; }
        h->maps_owned = true;
    }
    for (int i = 0; i < 2; i++) {
        if (h->d_map1[i]) { H_HIP(h, hipFree(h->d_map1[i])); h->d_map1[i] = nullptr; }
        if (h->d_map2[i]) { H_HIP(h, hipFree(h->d_map2[i])); h->d_map2[i] = nullptr; }
        const GipPinhole& c = i == 0 ? h->model.left : h->model.right;
        H_HIP(h, hipMalloc((void**)&h->d_map1[i], n * 4));
        H_HIP(h, hipMalloc((void**)&h->d_map2[i], n * 2));
        H_HIP(h, hipMemcpy(h->d_map1[i], c.map1.data(), n * 4, hipMemcpyHostToDevice));
        H_HIP(h, hipMemcpy(h->d_map2[i], c.map2.data(), n * 2, hipMemcpyHostToDevice));
    }
    h->map_w = w;
    h->map_h = hh;
    return GIP_OK;
}

void host_send_cb(void* p)
{
    Sender* s = (Sender*)p;
    // GPUSenderIfc::enqueueSend lambda: fillInData(); publish(); data_sent_ = true  (GpuSenderIfc.cpp:16-25)
    if (s->kind == SEND_IMAGE && s->icb) s->icb(&s->img, s->user);
    if (s->kind == SEND_DISPARITY && s->dcb) s->dcb(&s->dmsg, s->user);
    if (s->kind == SEND_POINTS && s->pcb) s->pcb(&s->pc, s->user);
    s->sent.store(true, std::memory_order_release);
}

}  // namespace

extern "C" {

const char* gip_status_string(int status)
{
    switch (status) {
    case GIP_OK: return "ok";
    case GIP_ERR_INVALID_ARG: return "invalid argument";
    case GIP_ERR_NOT_INITIALISED: return "stereo model not initialised";
    case GIP_ERR_HIP: return "HIP runtime error";
    case GIP_ERR_UNSUPPORTED: return "unsupported configuration";
    case GIP_ERR_NO_DEVICE: return "no HIP device";
    case GIP_ERR_EMPTY_SLOT: return "buffer slot is empty";
    case GIP_ERR_IO: return "I/O error";
    default: return "unknown status";
    }
}

int gip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// GpuStereoProcessor::GpuStereoProcessor (GPUStereoProcessor.cpp:12-39): createStereoBM(48, 19), XSOBEL, prefilter
// size 5, and the cuda getters' constants copied into the CPU matcher (texture 3, uniqueness 0, disp12MaxDiff 0).
int gip_create(int device, gip_handle* out)
{
    if (!out) return GIP_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return GIP_ERR_NO_DEVICE;
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) return GIP_ERR_NO_DEVICE;
    }
    if (device >= n) return GIP_ERR_INVALID_ARG;
    if (hipSetDevice(device) != hipSuccess) return GIP_ERR_HIP;
    gip_processor* h = new (std::nothrow) gip_processor();
    if (!h) return GIP_ERR_HIP;
    h->device = device;
    gip_bm_params& p = h->params;
    p.prefilter_type = GIP_PREFILTER_XSOBEL;
    p.prefilter_size = 5;
    p.prefilter_cap = 31;
    p.block_size = 19;
    p.min_disparity = 0;
    p.num_disparities = 48;
    p.texture_threshold = 3;
    p.uniqueness_ratio = 0;
    p.disp12_max_diff = 0;
    p.speckle_window_size = 0;
    p.speckle_range = 0;
    *out = h;
    return GIP_OK;
}

int gip_destroy(gip_handle h)
{
    if (!h) return GIP_ERR_INVALID_ARG;
    (void)hipSetDevice(h->device);        // teardown: errors have nowhere to go, every resource is released regardless
    if (h->l_strm) (void)hipStreamSynchronize(h->l_strm);
    if (h->r_strm) (void)hipStreamSynchronize(h->r_strm);
    (void)gip_clean_senders(h);
    for (auto& kv : h->slots) {
        Slot& s = kv.second;
        if (s.d) (void)hipFree(s.d);
        for (int k = 0; k < 2; k++) {
            if (s.stage[k]) (void)hipHostFree(s.stage[k]);
            if (s.stage_done[k]) (void)hipEventDestroy(s.stage_done[k]);
        }
        if (s.ready) (void)hipEventDestroy(s.ready);
        if (s.read) (void)hipEventDestroy(s.read);
    }
    for (auto& pp : h->pinned_pool) (void)hipHostFree(pp.first);
    if (h->maps_owned)
        for (int i = 0; i < 2; i++) {
            if (h->d_map1[i]) (void)hipFree(h->d_map1[i]);
            if (h->d_map2[i]) (void)hipFree(h->d_map2[i]);
        }
    DevBuf* bufs[] = {&h->bm_scratch, &h->spk_scratch, &h->pc_dev, &h->b_mono[0], &h->b_mono[1], &h->b_color, &h->b_rect[0], &h->b_rect[1], &h->b_rect_color, &h->b_f32, &h->w_tab};
    for (DevBuf* b : bufs)
        if (b->p) (void)hipFree(b->p);
    for (hipEvent_t e : h->tev)
        if (e) (void)hipEventDestroy(e);
    if (h->match_fork) (void)hipEventDestroy(h->match_fork);
    if (h->match_join) (void)hipEventDestroy(h->match_join);
    if (h->back_done) (void)hipEventDestroy(h->back_done);
    if (h->l_strm) (void)hipStreamDestroy(h->l_strm);
    if (h->r_strm) (void)hipStreamDestroy(h->r_strm);
    delete h;
    return GIP_OK;
}

const char* gip_last_error(gip_handle h) { return h ? h->err.c_str() : "null handle"; }
int gip_device(gip_handle h) { return h ? h->device : -1; }

// ------------------------------------------------------------------------------------------------ model
int gip_init_stereo_model(gip_handle h, const gip_camera_info* left, const gip_camera_info* right)
{
    H_CHECK(h);
    if (!left || !right) return fail(h, GIP_ERR_INVALID_ARG, "null CameraInfo");
    H_HIP(h, hipSetDevice(h->device));
    int rc = gip_model_init(h->model, *left, *right);
    if (rc != GIP_OK) return fail(h, rc, "invalid CameraInfo (size mismatch, bad distortion vector or zero fx)");
    // the reference never sets r_cam_name (GPUStereoProcessor.cpp:44-45 assigns l_cam_name_ twice); both are set here
    if (h->l_cam_name.empty()) h->l_cam_name = "left";
    if (h->r_cam_name.empty()) h->r_cam_name = "right";
    return upload_maps(h);
}

int gip_read_calibration(const char* file, gip_camera_info* out, char* camera_name, size_t name_len)
{
    if (!file || !out) return GIP_ERR_INVALID_ARG;
    std::string name, err;
    int rc = gip_parse_calibration_yaml(file, *out, name, err);
    if (rc == GIP_OK && camera_name && name_len) copy_cstr(camera_name, name_len, name.c_str());
    return rc;
}

int gip_init_stereo_model_files(gip_handle h, const char* left_cal_file, const char* right_cal_file)
{
    H_CHECK(h);
    if (!left_cal_file || !right_cal_file) return fail(h, GIP_ERR_INVALID_ARG, "null calibration path");
    gip_camera_info l, r;
    std::string err;
    int rc = gip_parse_calibration_yaml(left_cal_file, l, h->l_cam_name, err);
    if (rc != GIP_OK) return fail(h, rc, err);
    rc = gip_parse_calibration_yaml(right_cal_file, r, h->r_cam_name, err);
    if (rc != GIP_OK) return fail(h, rc, err);
    return gip_init_stereo_model(h, &l, &r);
}

int gip_is_stereo_model_initialised(gip_handle h) { return h && h->model.initialised ? 1 : 0; }

int gip_build_rectify_maps(const gip_camera_info* info, int16_t* map1, uint16_t* map2)
{
    if (!info || !map1 || !map2 || info->width <= 0 || info->height <= 0 || info->width > 32767 || info->height > 32767 || info->nD < 0 || info->nD > 8)
        return GIP_ERR_INVALID_ARG;
    gip_camera_info eff;
    if (!gip_effective_camera(*info, eff)) return GIP_ERR_INVALID_ARG;   // the tables have eff.width x eff.height entries
    std::vector<int16_t> m1;
    std::vector<uint16_t> m2;
    gip_build_rectify_map(*info, m1, m2);
    std::memcpy(map1, m1.data(), m1.size() * sizeof(int16_t));
    std::memcpy(map2, m2.data(), m2.size() * sizeof(uint16_t));
    return GIP_OK;
}

int gip_model_size(gip_handle h, int* width, int* height)
{
    H_CHECK(h);
    if (!h->model.initialised) return fail(h, GIP_ERR_NOT_INITIALISED, "stereo model not initialised");
    if (width) *width = h->model.left.info.width;
    if (height) *height = h->model.left.info.height;
    return GIP_OK;
}

int gip_model_get(gip_handle h, double* fx, double* fy, double* cx_l, double* cy, double* cx_r, double* baseline, double Q[16])
{
    H_CHECK(h);
    if (!h->model.initialised) return fail(h, GIP_ERR_NOT_INITIALISED, "stereo model not initialised");
    if (fx) *fx = h->model.left.fx();
    if (fy) *fy = h->model.left.fy();
    if (cx_l) *cx_l = h->model.left.cx();
    if (cy) *cy = h->model.left.cy();
    if (cx_r) *cx_r = h->model.right.cx();
    if (baseline) *baseline = h->model.baseline();
    if (Q) std::memcpy(Q, h->model.Q, sizeof h->model.Q);
    return GIP_OK;
}

int gip_get_rectify_maps(gip_handle h, int side, int16_t* map1, uint16_t* map2)
{
    H_CHECK(h);
    if (!h->model.initialised) return fail(h, GIP_ERR_NOT_INITIALISED, "stereo model not initialised");
    const GipPinhole& c = (side & GIP_SIDE_L) ? h->model.left : h->model.right;
    if (map1) std::memcpy(map1, c.map1.data(), c.map1.size() * sizeof(int16_t));
    if (map2) std::memcpy(map2, c.map2.data(), c.map2.size() * sizeof(uint16_t));
    return GIP_OK;
}

// ------------------------------------------------------------------------------------------------ upload / download
}  // extern "C"

int gip_handle_view(gip_handle h, GipHandleView* v)
{
    if (!h || !v) return GIP_ERR_INVALID_ARG;
    if (!h->model.initialised || !h->d_map1[0]) return fail(h, GIP_ERR_NOT_INITIALISED, "stereo model not initialised");
    v->device = h->device;
    v->width = h->map_w;
    v->height = h->map_h;
    for (int i = 0; i < 2; i++) {
        v->map1[i] = h->d_map1[i];
        v->map2[i] = h->d_map2[i];
    }
    v->distorted[0] = h->model.left.distorted;
    v->distorted[1] = h->model.right.distorted;
    std::memcpy(v->Q, h->model.Q, sizeof v->Q);
    v->cx_l = h->model.left.cx();
    v->cx_r = h->model.right.cx();
    v->params = h->params;
    v->max_speckle_size = h->max_speckle_size;
    v->max_speckle_diff = h->max_speckle_diff;
    return GIP_OK;
}

int gip_handle_clone(gip_handle src, gip_handle* out)
{
    if (!src || !out) return GIP_ERR_INVALID_ARG;
    *out = nullptr;
    if (!src->model.initialised || !src->d_map1[0]) return fail(src, GIP_ERR_NOT_INITIALISED, "stereo model not initialised");
    gip_handle h = nullptr;
    int rc = gip_create(src->device, &h);
    if (rc != GIP_OK) return rc;
    h->model = src->model;
    h->params = src->params;
    h->max_speckle_size = src->max_speckle_size;
    h->max_speckle_diff = src->max_speckle_diff;
    h->refine_disparity = src->refine_disparity;
    h->l_cam_name = src->l_cam_name;
    h->r_cam_name = src->r_cam_name;
    for (int i = 0; i < 2; i++) {
        h->d_map1[i] = src->d_map1[i];
        h->d_map2[i] = src->d_map2[i];
    }
    h->maps_owned = false;
    h->map_w = src->map_w;
    h->map_h = src->map_h;
    // (a clone only ever runs gip_batch_run on the caller's streams: its side streams are never created, stream_of)
    *out = h;
    return GIP_OK;
}

extern "C" {

int gip_upload_mat(gip_handle h, int slot, const void* data, int rows, int cols, size_t step, int type, const char* encoding)
{
    H_CHECK(h);
    const int es = elem_size(type);
    if (!data || rows <= 0 || cols <= 0 || es <= 0) return fail(h, GIP_ERR_INVALID_ARG, "uploadMat: bad image");
    const size_t row_bytes = (size_t)cols * es;
    if (step == 0) step = row_bytes;
    if (step < row_bytes) return fail(h, GIP_ERR_INVALID_ARG, "uploadMat: step smaller than a row");
    H_HIP(h, hipSetDevice(h->device));
    Slot* s;
    int rc = slot_create(h, slot, rows, cols, type, &s);
    if (rc != GIP_OK) return rc;
    hipStream_t st = stream_of(h, slot);
    // pinned staging: the caller's buffer is free again when this returns (as with the reference's CPU copy)
    const size_t bytes = row_bytes * rows;
    const int k = s->stage_next;            // the buffer whose H2D copy was issued two uploads ago
    s->stage_next ^= 1;
    if (s->stage_busy[k]) {
        H_HIP(h, hipEventSynchronize(s->stage_done[k]));
        s->stage_busy[k] = false;
    }
    if (s->stage_cap[k] < bytes) {
        if (s->stage[k]) H_HIP(h, hipHostFree(s->stage[k]));
        s->stage[k] = nullptr;
        H_HIP(h, hipHostMalloc(&s->stage[k], bytes, hipHostMallocDefault));
        s->stage_cap[k] = bytes;
    }
    if (step == row_bytes) std::memcpy(s->stage[k], data, bytes);
    else
        for (int y = 0; y < rows; y++) std::memcpy((uint8_t*)s->stage[k] + (size_t)y * row_bytes, (const uint8_t*)data + (size_t)y * step, row_bytes);
    rc = begin_write(h, s, st);
    if (rc != GIP_OK) return rc;
    H_HIP(h, hipMemcpyAsync(s->d, s->stage[k], bytes, hipMemcpyHostToDevice, st));   // s->pitch == row_bytes
    H_HIP(h, hipEventRecord(s->stage_done[k], st));
    s->stage_busy[k] = true;
    rc = end_write(h, s, st);
    if (rc != GIP_OK) return rc;
    if (encoding && encoding[0]) {
        // store the encoding per side (GPUStereoProcessor.cpp:100-108)
        if ((slot & GIP_SIDE_MASK) == GIP_SIDE_L) h->l_enc = encoding;
        else h->r_enc = encoding;
    }
    return GIP_OK;
}

int gip_mat_info(gip_handle h, int slot, int* rows, int* cols, int* type)
{
    H_CHECK(h);
    Slot* s = slot_get(h, slot);
    if (!s) return fail(h, GIP_ERR_EMPTY_SLOT, "slot is empty");
    if (rows) *rows = s->rows;
    if (cols) *cols = s->cols;
    if (type) *type = s->type;
    return GIP_OK;
}

int gip_mat_device_ptr(gip_handle h, int slot, void** dptr, size_t* pitch)
{
    H_CHECK(h);
    Slot* s = slot_get(h, slot);
    if (dptr) *dptr = s ? s->d : nullptr;
    if (pitch) *pitch = s ? s->pitch : 0;
    return s ? GIP_OK : GIP_ERR_EMPTY_SLOT;
}

int gip_download_mat(gip_handle h, int slot, void* dst, size_t dst_step)
{
    H_CHECK(h);
    Slot* s = slot_get(h, slot);
    if (!s) return fail(h, GIP_ERR_EMPTY_SLOT, "downloadMat: slot is empty");
    if (!dst) return fail(h, GIP_ERR_INVALID_ARG, "downloadMat: null destination");
    H_HIP(h, hipSetDevice(h->device));
    const size_t row_bytes = (size_t)s->cols * elem_size(s->type);
    if (dst_step == 0) dst_step = row_bytes;
    if (dst_step < row_bytes) return fail(h, GIP_ERR_INVALID_ARG, "downloadMat: step smaller than a row");
    hipStream_t st = stream_of(h, slot);
    int rc = begin_read(h, s, st);
    if (rc != GIP_OK) return rc;
    if (dst_step == row_bytes && s->pitch == row_bytes)
        H_HIP(h, hipMemcpyAsync(dst, s->d, row_bytes * s->rows, hipMemcpyDeviceToHost, st));
    else
        H_HIP(h, hipMemcpy2DAsync(dst, dst_step, s->d, s->pitch, row_bytes, s->rows, hipMemcpyDeviceToHost, st));
    rc = end_read(h, s, st);
    if (rc != GIP_OK) return rc;
    H_HIP(h, hipStreamSynchronize(st));  // getStream(mat_source).waitForCompletion()  (GPUStereoProcessor.cpp:115)
    return GIP_OK;
}

// ------------------------------------------------------------------------------------------------ colour
int gip_convert_color(gip_handle h, int src_slot, int dst_slot, const char* src_encoding, const char* dst_encoding)
{
    H_CHECK(h);
    if (!src_encoding || !src_encoding[0] || !dst_encoding || !dst_encoding[0])
        return fail(h, GIP_ERR_INVALID_ARG, "convertColor: empty encoding (reference asserts, GPUStereoProcessor.cpp:121-122)");
    Slot* src = slot_get(h, src_slot);
    if (!src) return fail(h, GIP_ERR_EMPTY_SLOT, "convertColor: source slot is empty");
    H_HIP(h, hipSetDevice(h->device));
    const std::string se = src_encoding, de = dst_encoding;
    hipStream_t st = stream_of(h, src_slot);  // the reference converts on the SOURCE slot's stream (:126)
    const int st_type = encoding_type(se), dt_type = encoding_type(de);
    if (st_type < 0 || dt_type < 0) return fail(h, GIP_ERR_UNSUPPORTED, "convertColor: unsupported encoding " + se + " -> " + de);
    if (st_type != src->type) return fail(h, GIP_ERR_INVALID_ARG, "convertColor: slot type does not match encoding " + se);
    const int rows = src->rows, cols = src->cols;
    const void* sd = src->d;
    const size_t sp = src->pitch;
    Slot* dst;
    int rc = slot_create(h, dst_slot, rows, cols, se == de ? src->type : dt_type, &dst);
    if (rc != GIP_OK) return rc;
    src = slot_get(h, src_slot);  // the map may have rehashed
    if ((rc = begin_read(h, src, st)) != GIP_OK) return rc;
    if ((rc = begin_write(h, dst, st)) != GIP_OK) return rc;
    const int pat = bayer_pattern(se);
    if (se == de) {
        H_HIP(h, hipMemcpyAsync(dst->d, sd, (size_t)cols * elem_size(src->type) * rows, hipMemcpyDeviceToDevice, st));   // tight pitches
    } else if (pat >= 0 && de == "mono8") {
        rc = gipk_cvt_bayer_to_gray((const uint8_t*)sd, sp, (uint8_t*)dst->d, dst->pitch, cols, rows, pat, 1, 0, 0, st);
    } else if (pat >= 0 && de == "bgr8") {
        rc = gipk_cvt_bayer_to_bgr((const uint8_t*)sd, sp, (uint8_t*)dst->d, dst->pitch, cols, rows, pat, 1, 0, 0, st);
    } else if ((se == "bgr8" || se == "rgb8") && de == "mono8") {
        rc = gipk_cvt_bgr_to_gray((const uint8_t*)sd, sp, (uint8_t*)dst->d, dst->pitch, cols, rows, se == "rgb8", 1, 0, 0, st);
    } else if (se == "rgb8" && de == "bgr8") {
        rc = gipk_cvt_swap_rb((const uint8_t*)sd, sp, (uint8_t*)dst->d, dst->pitch, cols, rows, 1, 0, 0, st);
    } else if (se == "bgr8" && de == "rgb8") {
        rc = gipk_cvt_swap_rb((const uint8_t*)sd, sp, (uint8_t*)dst->d, dst->pitch, cols, rows, 1, 0, 0, st);
    } else if (se == "mono8" && de == "bgr8") {
        rc = gipk_cvt_gray_to_bgr((const uint8_t*)sd, sp, (uint8_t*)dst->d, dst->pitch, cols, rows, 1, 0, 0, st);
    } else if (se == "mono8" && de == "mono16") {
        rc = gipk_cvt_mono8_to_mono16((const uint8_t*)sd, sp, (uint16_t*)dst->d, dst->pitch, cols, rows, 1, 0, 0, st);
    } else if (se == "mono16" && de == "mono8") {
        rc = gipk_cvt_mono16_to_mono8((const uint16_t*)sd, sp, (uint8_t*)dst->d, dst->pitch, cols, rows, 1, 0, 0, st);
    } else {
        return fail(h, GIP_ERR_UNSUPPORTED, "convertColor: no single conversion " + se + " -> " + de + " (reference asserts one code, GPUStereoProcessor.cpp:138)");
    }
    if (rc != GIP_OK) return propagate(h, rc, "convertColor");
    if ((rc = end_read(h, src, st)) != GIP_OK) return rc;
    return end_write(h, dst, st);
}

int gip_convert_raw_to_color(gip_handle h, int side)
{
    H_CHECK(h);
    const bool left = (side & GIP_SIDE_MASK) == GIP_SIDE_L;
    return gip_convert_color(h, (left ? GIP_SIDE_L : GIP_SIDE_R) | GIP_SRC_RAW, (left ? GIP_SIDE_L : GIP_SIDE_R) | GIP_SRC_COLOR,
                             (left ? h->l_enc : h->r_enc).c_str(), "bgr8");
}

int gip_convert_raw_to_mono(gip_handle h, int side)
{
    H_CHECK(h);
    const bool left = (side & GIP_SIDE_MASK) == GIP_SIDE_L;
    return gip_convert_color(h, (left ? GIP_SIDE_L : GIP_SIDE_R) | GIP_SRC_RAW, (left ? GIP_SIDE_L : GIP_SIDE_R) | GIP_SRC_MONO,
                             (left ? h->l_enc : h->r_enc).c_str(), "mono8");
}

// ------------------------------------------------------------------------------------------------ rectify
int gip_rectify_image(gip_handle h, int src_slot, int dst_slot, int interpolation)
{
    H_CHECK(h);
    if (!h->model.initialised) return fail(h, GIP_ERR_NOT_INITIALISED, "rectifyImage: stereo model not initialised");
    Slot* src = slot_get(h, src_slot);
    if (!src) return fail(h, GIP_ERR_EMPTY_SLOT, "rectifyImage: source slot is empty");
    if (interpolation != GIP_INTER_LINEAR && interpolation != GIP_INTER_NEAREST)
        return fail(h, GIP_ERR_UNSUPPORTED, "rectifyImage: only INTER_LINEAR and INTER_NEAREST are implemented");
    if (src->type != GIP_8UC1 && src->type != GIP_8UC3) return fail(h, GIP_ERR_UNSUPPORTED, "rectifyImage: 8UC1 / 8UC3 only");
    if (src->cols != h->map_w || src->rows != h->map_h) return fail(h, GIP_ERR_INVALID_ARG, "rectifyImage: image size differs from the calibration");
    H_HIP(h, hipSetDevice(h->device));
    const bool left = (src_slot & GIP_SIDE_L) == GIP_SIDE_L;  // side of the SOURCE selects camera and stream (:241-249)
    hipStream_t st = stream_of(h, src_slot);
    const int rows = src->rows, cols = src->cols, type = src->type;
    Slot* dst;
    int rc = slot_create(h, dst_slot, rows, cols, type, &dst);
    if (rc != GIP_OK) return rc;
    src = slot_get(h, src_slot);
    if ((rc = begin_read(h, src, st)) != GIP_OK) return rc;
    if ((rc = begin_write(h, dst, st)) != GIP_OK) return rc;
    const GipPinhole& cam = left ? h->model.left : h->model.right;
    const int cn = (type >> 3) + 1;
    if (!cam.distorted) {
        // image_geometry: distortion state NONE -> raw.copyTo(rectified)
        H_HIP(h, hipMemcpyAsync(dst->d, src->d, (size_t)cols * cn * rows, hipMemcpyDeviceToDevice, st));   // tight pitches
    } else {
        const int i = left ? 0 : 1;
        rc = gipk_remap_u8_interp((const uint8_t*)src->d, src->pitch, cols, rows, (uint8_t*)dst->d, dst->pitch, cols, rows, cn, h->d_map1[i], h->d_map2[i], interpolation, 1, 0,
                                  0, st);
        if (rc != GIP_OK) return propagate(h, rc, "rectifyImage");
    }
    if ((rc = end_read(h, src, st)) != GIP_OK) return rc;
    return end_write(h, dst, st);
}

// ------------------------------------------------------------------------------------------------ disparity
static int refresh_f32(gip_handle h, int disparity_slot, hipStream_t st)
{
    Slot* d = slot_get(h, disparity_slot);
    if (!d) return GIP_ERR_EMPTY_SLOT;
    const int rows = d->rows, cols = d->cols;
    const int f32_slot = (disparity_slot & GIP_SIDE_MASK) | GIP_SRC_DISPARITY_32F;
    Slot* f;
    int rc = slot_create(h, f32_slot, rows, cols, GIP_32FC1, &f);
    if (rc != GIP_OK) return rc;
    d = slot_get(h, disparity_slot);
    if ((rc = begin_read(h, d, st)) != GIP_OK) return rc;
    if ((rc = begin_write(h, f, st)) != GIP_OK) return rc;
    rc = gipk_disparity_to_f32((const int16_t*)d->d, d->pitch, (float*)f->d, f->pitch, cols, rows, h->model.left.cx(), h->model.right.cx(), 1, 0, 0, st);
    if (rc != GIP_OK) return propagate(h, rc, "disparity -> float");
    if ((rc = end_read(h, d, st)) != GIP_OK) return rc;
    return end_write(h, f, st);
}

int gip_compute_disparity(gip_handle h, int left_slot, int right_slot, int disparity_slot)
{
    H_CHECK(h);
    if (!h->model.initialised) return fail(h, GIP_ERR_NOT_INITIALISED, "computeDisparity: stereo model not initialised");
    // "the x offset will be different for right side ... assert" (GPUStereoProcessor.cpp:286-288)
    if ((disparity_slot & GIP_SIDE_MASK) != GIP_SIDE_L) return fail(h, GIP_ERR_UNSUPPORTED, "computeDisparity: only the left disparity is supported");
    Slot* l = slot_get(h, left_slot);
    Slot* r = slot_get(h, right_slot);
    if (!l || !r) return fail(h, GIP_ERR_EMPTY_SLOT, "computeDisparity: input slot is empty");
    if (l->type != GIP_8UC1 || r->type != GIP_8UC1 || l->rows != r->rows || l->cols != r->cols)
        return fail(h, GIP_ERR_INVALID_ARG, "computeDisparity: inputs must be equal-size 8UC1");
    H_HIP(h, hipSetDevice(h->device));
    const int rows = l->rows, cols = l->cols;
    int rc = gip_validate_bm_params(&h->params, cols, rows);
    if (rc != GIP_OK) return fail(h, rc, "computeDisparity: StereoBM parameter validation failed");
    hipStream_t st = stream_of(h, disparity_slot);
    Slot* d;
    rc = slot_create(h, disparity_slot, rows, cols, GIP_16SC1, &d);
    if (rc != GIP_OK) return rc;
    l = slot_get(h, left_slot);
    r = slot_get(h, right_slot);
    if (l->pitch != r->pitch) return fail(h, GIP_ERR_INVALID_ARG, "computeDisparity: pitch mismatch");
    rc = ensure_dev(h, h->bm_scratch, gipk_stereobm_scratch_bytes(cols, rows, 1, &h->params));
    if (rc != GIP_OK) return rc;
    if ((rc = begin_read(h, l, st)) != GIP_OK) return rc;
    if ((rc = begin_read(h, r, st)) != GIP_OK) return rc;
    if ((rc = begin_write(h, d, st)) != GIP_OK) return rc;
    rc = gipk_stereobm((const uint8_t*)l->d, (const uint8_t*)r->d, l->pitch, cols, rows, &h->params, (int16_t*)d->d, d->pitch, h->bm_scratch.p,
                       h->bm_scratch.cap, 1, 0, 0, st);
    if (rc != GIP_OK) return propagate(h, rc, "computeDisparity");
    if ((rc = end_read(h, l, st)) != GIP_OK) return rc;
    if ((rc = end_read(h, r, st)) != GIP_OK) return rc;
    if ((rc = end_write(h, d, st)) != GIP_OK) return rc;
    // disparity.convertTo(CV_32FC1, 1/16, -(cx_l - cx_r))  (GPUStereoProcessor.cpp:320; commented out on the GPU path :293-295)
    return refresh_f32(h, disparity_slot, st);
}

int gip_filter_speckles(gip_handle h, int disparity_slot)
{
    H_CHECK(h);
    if (h->max_speckle_size <= 0) return GIP_OK;  // GPUStereoProcessor.cpp:358
    Slot* d = slot_get(h, disparity_slot);
    if (!d) return fail(h, GIP_ERR_EMPTY_SLOT, "filterSpeckles: slot is empty");
    if (d->type != GIP_16SC1) return fail(h, GIP_ERR_INVALID_ARG, "filterSpeckles: slot must hold a 16SC1 disparity");
    H_HIP(h, hipSetDevice(h->device));
    hipStream_t st = stream_of(h, disparity_slot);
    int rc = ensure_dev(h, h->spk_scratch, gipk_speckle_scratch_bytes(d->cols, d->rows, 1));
    if (rc != GIP_OK) return rc;
    if ((rc = begin_write(h, d, st)) != GIP_OK) return rc;
    // The reference filters its 8-bit integer map with newVal 0 and maxDiff in pixels (:383); this map is x16 fixed
    // point with FILTERED = (minDisparity-1)*16 as the invalid value, so newVal = FILTERED and maxDiff scales by 16.
    const int new_val = (h->params.min_disparity - 1) * 16;
    const int max_diff = (int)(h->max_speckle_diff * 16.0);
    rc = gipk_filter_speckles((int16_t*)d->d, d->pitch, d->cols, d->rows, new_val, h->max_speckle_size, max_diff, h->spk_scratch.p, h->spk_scratch.cap, 1, 0, st);
    if (rc != GIP_OK) return propagate(h, rc, "filterSpeckles");
    if ((rc = end_write(h, d, st)) != GIP_OK) return rc;
    if (h->model.initialised) return refresh_f32(h, disparity_slot, st);
    return GIP_OK;
}

int gip_compute_disparity_image(gip_handle h, int disparity_slot, int image_slot)
{
    H_CHECK(h);
    if (!h->model.initialised) return fail(h, GIP_ERR_NOT_INITIALISED, "computeDisparityImage: stereo model not initialised");
    Slot* d = slot_get(h, disparity_slot);
    if (!d || d->type != GIP_16SC1) return fail(h, GIP_ERR_EMPTY_SLOT, "computeDisparityImage: no 16SC1 disparity in the slot");
    H_HIP(h, hipSetDevice(h->device));
    hipStream_t st = stream_of(h, disparity_slot);
    const int rows = d->rows, cols = d->cols;
    Slot* img;
    int rc = slot_create(h, image_slot, rows, cols, GIP_8UC4, &img);
    if (rc != GIP_OK) return rc;
    d = slot_get(h, disparity_slot);
    if ((rc = begin_read(h, d, st)) != GIP_OK) return rc;
    if ((rc = begin_write(h, img, st)) != GIP_OK) return rc;
    rc = gipk_draw_color_disp((const int16_t*)d->d, d->pitch, (uint8_t*)img->d, img->pitch, cols, rows, h->params.num_disparities, 1, 0, 0, st);
    if (rc != GIP_OK) return propagate(h, rc, "computeDisparityImage");
    if ((rc = end_read(h, d, st)) != GIP_OK) return rc;
    return end_write(h, img, st);
}

int gip_project_disparity_to_3d_points(gip_handle h, int disparity_slot, int points_slot)
{
    H_CHECK(h);
    if (!h->model.initialised) return fail(h, GIP_ERR_NOT_INITIALISED, "projectDisparityTo3DPoints: stereo model not initialised");
    const int f32_slot = (disparity_slot & GIP_SIDE_MASK) | GIP_SRC_DISPARITY_32F;
    Slot* f = slot_get(h, (disparity_slot & GIP_SRC_DISPARITY_32F) ? disparity_slot : f32_slot);
    if (!f || f->type != GIP_32FC1) return fail(h, GIP_ERR_EMPTY_SLOT, "projectDisparityTo3DPoints: no float disparity (run computeDisparity first)");
    H_HIP(h, hipSetDevice(h->device));
    hipStream_t st = stream_of(h, disparity_slot);
    const int rows = f->rows, cols = f->cols;
    Slot* pts;
    int rc = slot_create(h, points_slot, rows, cols, GIP_32FC3, &pts);
    if (rc != GIP_OK) return rc;
    f = slot_get(h, (disparity_slot & GIP_SRC_DISPARITY_32F) ? disparity_slot : f32_slot);
    if ((rc = begin_read(h, f, st)) != GIP_OK) return rc;
    if ((rc = begin_write(h, pts, st)) != GIP_OK) return rc;
    // handleMissingValues = true: the image minimum is FILTERED, converted exactly like the map itself
    const float missing = (float)((h->params.min_disparity - 1) * 16) * (float)(1.0 / 16) + (float)(-(h->model.left.cx() - h->model.right.cx()));
    rc = gipk_reproject_to_3d((const float*)f->d, f->pitch, (float*)pts->d, pts->pitch, cols, rows, h->model.Q, missing, 1, 1, 0, 0, st);
    if (rc != GIP_OK) return propagate(h, rc, "projectDisparityTo3DPoints");
    if ((rc = end_read(h, f, st)) != GIP_OK) return rc;
    return end_write(h, pts, st);
}

// ------------------------------------------------------------------------------------------------ senders
static int new_sender(gip_handle h, SenderKind kind, size_t host_bytes, Sender** out, int* sender_id)
{
    std::unique_ptr<Sender> s(new Sender());
    s->kind = kind;
    s->host = pinned_acquire(h, host_bytes);
    if (!s->host) return fail(h, GIP_ERR_HIP, "hipHostMalloc failed for a sender payload");
    s->host_bytes = host_bytes;
    *out = s.get();
    if (sender_id) *sender_id = (int)h->senders.size();
    h->senders.push_back(std::move(s));
    return GIP_OK;
}

int gip_enqueue_send_image(gip_handle h, int slot, const gip_header* header, const char* encoding, gip_image_cb cb, void* user, int* sender_id)
{
    H_CHECK(h);
    Slot* s = slot_get(h, slot);
    if (!s) return fail(h, GIP_ERR_EMPTY_SLOT, "enqueueSendImage: slot is empty");
    if (!encoding) return fail(h, GIP_ERR_INVALID_ARG, "enqueueSendImage: null encoding");
    H_HIP(h, hipSetDevice(h->device));
    hipStream_t st = stream_of(h, slot);
    const int et = encoding_type(encoding);
    // step = width * bitDepth * channels / 8 of the ENCODING (GpuSenderImage.cpp:20-22)
    const size_t step = (size_t)s->cols * (et >= 0 ? elem_size(et) : elem_size(s->type));
    const size_t row_bytes = (size_t)s->cols * elem_size(s->type);
    if (step < row_bytes) return fail(h, GIP_ERR_INVALID_ARG, "enqueueSendImage: encoding narrower than the buffer");
    Sender* snd;
    int rc = new_sender(h, SEND_IMAGE, step * s->rows, &snd, sender_id);
    if (rc != GIP_OK) return rc;
    if (header) snd->img.header = *header;
    copy_cstr(snd->img.encoding, sizeof snd->img.encoding, encoding);
    snd->img.height = s->rows;
    snd->img.width = s->cols;
    snd->img.is_bigendian = 0;
    snd->img.step = (uint32_t)step;
    snd->img.data = (uint8_t*)snd->host;
    snd->img.data_size = step * s->rows;
    snd->icb = cb;
    snd->user = user;
    if ((rc = begin_read(h, s, st)) != GIP_OK) return rc;
    if (step == row_bytes && s->pitch == row_bytes)
        H_HIP(h, hipMemcpyAsync(snd->host, s->d, row_bytes * s->rows, hipMemcpyDeviceToHost, st));
    else
        H_HIP(h, hipMemcpy2DAsync(snd->host, step, s->d, s->pitch, row_bytes, s->rows, hipMemcpyDeviceToHost, st));
    if ((rc = end_read(h, s, st)) != GIP_OK) return rc;
    H_HIP(h, hipLaunchHostFunc(st, host_send_cb, snd));
    return GIP_OK;
}

int gip_enqueue_send_disparity(gip_handle h, int slot, const gip_header* header, gip_disparity_cb cb, void* user, int* sender_id)
{
    H_CHECK(h);
    if (!h->model.initialised) return fail(h, GIP_ERR_NOT_INITIALISED, "enqueueSendDisparity: stereo model not initialised");
    const int f32_slot = (slot & GIP_SIDE_MASK) | GIP_SRC_DISPARITY_32F;
    Slot* f = slot_get(h, f32_slot);
    if (!f || f->type != GIP_32FC1) return fail(h, GIP_ERR_EMPTY_SLOT, "enqueueSendDisparity: no float disparity");
    H_HIP(h, hipSetDevice(h->device));
    hipStream_t st = stream_of(h, slot);
    const size_t step = (size_t)f->cols * 4;
    Sender* snd;
    int rc = new_sender(h, SEND_DISPARITY, step * f->rows, &snd, sender_id);
    if (rc != GIP_OK) return rc;
    gip_disparity_msg& m = snd->dmsg;
    if (header) { m.header = *header; m.image.header = *header; }
    copy_cstr(m.image.encoding, sizeof m.image.encoding, "32FC1");
    m.image.height = f->rows;
    m.image.width = f->cols;
    m.image.is_bigendian = 0;
    m.image.step = (uint32_t)step;
    m.image.data = (uint8_t*)snd->host;
    m.image.data_size = step * f->rows;
    if ((rc = gip_disparity_info(h, f->cols, f->rows, &m)) != GIP_OK) return rc;
    snd->dcb = cb;
    snd->user = user;
    if ((rc = begin_read(h, f, st)) != GIP_OK) return rc;
    if (f->pitch == step)
        H_HIP(h, hipMemcpyAsync(snd->host, f->d, step * f->rows, hipMemcpyDeviceToHost, st));
    else
        H_HIP(h, hipMemcpy2DAsync(snd->host, step, f->d, f->pitch, step, f->rows, hipMemcpyDeviceToHost, st));
    if ((rc = end_read(h, f, st)) != GIP_OK) return rc;
    H_HIP(h, hipLaunchHostFunc(st, host_send_cb, snd));
    return GIP_OK;
}

// the metadata of a stereo_msgs/DisparityImage for a width x height map computed with the handle's parameters and model
int gip_disparity_info(gip_handle h, int width, int height, gip_disparity_msg* out)
{
    H_CHECK(h);
    if (!out || width <= 0 || height <= 0) return fail(h, GIP_ERR_INVALID_ARG, "disparity_info: bad arguments");
    if (!h->model.initialised) return fail(h, GIP_ERR_NOT_INITIALISED, "disparity_info: stereo model not initialised");
    gip_disparity_msg& m = *out;
    // valid window / range / f / T as GPUSenderDisparity::fillInData computes them (GpuSenderDisparity.cpp:29-43), with
    // the constructor arguments in their DECLARED order (minDisparity, fx, baseline); the reference call site passes
    // (fx, baseline, minDisparity) into those positions (GPUStereoProcessor.cpp:221-222), which is a defect.
    const int block = h->params.block_size, ndisp = h->params.num_disparities, mind = h->params.min_disparity;
    const int border = block / 2;
    const int left = ndisp + mind + border - 1;
    const int wtf = (mind >= 0) ? border + mind : (border > -mind ? border : -mind);
    const int right = width - 1 - wtf;
    const int top = border, bottom = height - 1 - border;
    m.valid_x_offset = left;
    m.valid_y_offset = top;
    m.valid_width = right - left;
    m.valid_height = bottom - top;
    m.min_disparity = (float)(mind + 1);
    m.max_disparity = (float)(mind + ndisp - 1);
    m.f = (float)h->model.right.fx();
    m.T = (float)h->model.baseline();
    m.delta_d = 1.0f / 16;  // never set by the reference; stereo_image_proc's value for a x16 fixed-point matcher
    return GIP_OK;
}

int gip_enqueue_send_points(gip_handle h, int points_slot, int color_slot, const gip_header* header, gip_pointcloud2_cb cb, void* user, int* sender_id)
{
    H_CHECK(h);
    Slot* pts = slot_get(h, points_slot);
    Slot* col = slot_get(h, color_slot);
    if (!pts || !col) return fail(h, GIP_ERR_EMPTY_SLOT, "enqueueSendPoints: slot is empty");
    if (pts->type != GIP_32FC3 || col->type != GIP_8UC3 || pts->rows != col->rows || pts->cols != col->cols)
        return fail(h, GIP_ERR_INVALID_ARG, "enqueueSendPoints: need 32FC3 points and an equal-size 8UC3 colour image");
    H_HIP(h, hipSetDevice(h->device));
    hipStream_t st = stream_of(h, points_slot);
    const size_t bytes = (size_t)pts->rows * pts->cols * 32;
    int rc = ensure_dev(h, h->pc_dev, bytes);
    if (rc != GIP_OK) return rc;
    Sender* snd;
    rc = new_sender(h, SEND_POINTS, bytes, &snd, sender_id);
    if (rc != GIP_OK) return rc;
    gip_pointcloud2_msg& m = snd->pc;
    if (header) m.header = *header;
    m.height = pts->rows;
    m.width = pts->cols;
    m.is_bigendian = 0;
    m.is_dense = 0;
    // setPointCloud2FieldsByString(2, "xyz", "rgb")  (GpuSenderPc2.cpp:31-32)
    const char* names[4] = {"x", "y", "z", "rgb"};
    const uint32_t offs[4] = {0, 4, 8, 16};
    for (int i = 0; i < 4; i++) {
        copy_cstr(m.fields[i].name, sizeof m.fields[i].name, names[i]);
        m.fields[i].offset = offs[i];
        m.fields[i].datatype = 7;
        m.fields[i].count = 1;
    }
    m.nfields = 4;
    m.point_step = 32;
    m.row_step = 32 * m.width;
    m.data = (uint8_t*)snd->host;
    m.data_size = bytes;
    snd->pcb = cb;
    snd->user = user;
    if ((rc = begin_read(h, pts, st)) != GIP_OK) return rc;
    if ((rc = begin_read(h, col, st)) != GIP_OK) return rc;
    rc = gipk_pack_pointcloud2((const float*)pts->d, pts->pitch, (const uint8_t*)col->d, col->pitch, (uint8_t*)h->pc_dev.p, pts->cols, pts->rows, 1, 0, 0, 0, st);
    if (rc != GIP_OK) return propagate(h, rc, "enqueueSendPoints");
    if ((rc = end_read(h, pts, st)) != GIP_OK) return rc;
    if ((rc = end_read(h, col, st)) != GIP_OK) return rc;
    H_HIP(h, hipMemcpyAsync(snd->host, h->pc_dev.p, bytes, hipMemcpyDeviceToHost, st));
    H_HIP(h, hipLaunchHostFunc(st, host_send_cb, snd));
    return GIP_OK;
}

int gip_sender_was_sent(gip_handle h, int sender_id)
{
    if (!h || sender_id < 0 || sender_id >= (int)h->senders.size()) return 0;
    return h->senders[sender_id]->sent.load(std::memory_order_acquire) ? 1 : 0;
}

const gip_image_msg* gip_sender_image_msg(gip_handle h, int id)
{
    if (!h || id < 0 || id >= (int)h->senders.size() || h->senders[id]->kind != SEND_IMAGE) return nullptr;
    return &h->senders[id]->img;
}
const gip_disparity_msg* gip_sender_disparity_msg(gip_handle h, int id)
{
    if (!h || id < 0 || id >= (int)h->senders.size() || h->senders[id]->kind != SEND_DISPARITY) return nullptr;
    return &h->senders[id]->dmsg;
}
const gip_pointcloud2_msg* gip_sender_pointcloud2_msg(gip_handle h, int id)
{
    if (!h || id < 0 || id >= (int)h->senders.size() || h->senders[id]->kind != SEND_POINTS) return nullptr;
    return &h->senders[id]->pc;
}

int gip_clean_senders(gip_handle h)
{
    H_CHECK(h);
    if (h->senders.empty()) return GIP_OK;
    // a sender's payload may still be the target of an in-flight copy: drain first (the reference relies on
    // waitForAllStreams() before cleanSenders(), StereoProcessor.cpp:284-285)
    H_HIP(h, hipSetDevice(h->device));
    if (h->l_strm) H_HIP(h, hipStreamSynchronize(h->l_strm));
    if (h->r_strm) H_HIP(h, hipStreamSynchronize(h->r_strm));
    for (auto& s : h->senders)
        if (s->host) h->pinned_pool.emplace_back(s->host, s->host_bytes);
    h->senders.clear();
    return GIP_OK;
}

// ------------------------------------------------------------------------------------------------ sync + setters
int gip_wait_for_stream(gip_handle h, int slot_or_side)
{
    H_CHECK(h);
    H_HIP(h, hipSetDevice(h->device));
    H_HIP(h, hipStreamSynchronize(stream_of(h, slot_or_side)));
    return GIP_OK;
}

int gip_wait_for_all_streams(gip_handle h)
{
    H_CHECK(h);
    H_HIP(h, hipSetDevice(h->device));
    if (h->l_strm) H_HIP(h, hipStreamSynchronize(h->l_strm));
    if (h->r_strm) H_HIP(h, hipStreamSynchronize(h->r_strm));
    return GIP_OK;
}

// Setters validate what can never be valid (the reference forwards to cv::StereoBM setters that accept anything and fails later
// inside compute(); its configCb makes the window odd and the range a multiple of 16 first, reference src/StereoProcessor.cpp:309-311 --
// StereoProcessor::configCb of include/gpuimageproc/StereoProcessor.hpp does the same before calling these).  A rejected value
// leaves the parameter unchanged.
#define GIP_SETTER(cond, text)                                   \
    do {                                                         \
        H_CHECK(h);                                              \
        if (!(cond)) return fail(h, GIP_ERR_INVALID_ARG, text);  \
    } while (0)
int gip_set_pre_filter_type(gip_handle h, int v)
{
    GIP_SETTER(v == GIP_PREFILTER_NORMALIZED_RESPONSE || v == GIP_PREFILTER_XSOBEL, "setPreFilterType: 0 (NORMALIZED_RESPONSE) or 1 (XSOBEL)");
    h->params.prefilter_type = v;
    return GIP_OK;
}
int gip_set_refine_disparity(gip_handle h, int v) { H_CHECK(h); h->refine_disparity = v; return GIP_OK; }
int gip_set_block_size(gip_handle h, int v)
{
    GIP_SETTER(v >= 5 && v <= 255 && (v & 1), "setBlockSize: odd, 5..255 (cfg/GPU.cfg:16)");
    h->params.block_size = v;
    return GIP_OK;
}
int gip_set_num_disparities(gip_handle h, int v)
{
    GIP_SETTER(v > 0 && v % 16 == 0 && v <= 4096, "setNumDisparities: positive multiple of 16");
    h->params.num_disparities = v;
    return GIP_OK;
}
int gip_set_min_disparity(gip_handle h, int v)
{
    GIP_SETTER(v >= -2047 && v <= 2047, "setMinDisparity: |minDisparity| <= 2047 (x16 fixed point in int16)");
    h->params.min_disparity = v;
    return GIP_OK;
}
int gip_set_texture_threshold(gip_handle h, int v)
{
    GIP_SETTER(v >= 0, "setTextureThreshold: >= 0");
    h->params.texture_threshold = v;
    return GIP_OK;
}
int gip_set_max_speckle_size(gip_handle h, int v)
{
    GIP_SETTER(v >= 0, "setMaxSpeckleSize: >= 0 (0 disables the filter)");
    h->max_speckle_size = v;
    return GIP_OK;
}
int gip_get_max_speckle_size(gip_handle h) { return h ? h->max_speckle_size : 0; }
int gip_set_max_speckle_diff(gip_handle h, double v)
{
    GIP_SETTER(v >= 0.0 && v <= 2047.0, "setMaxSpeckleDiff: 0..2047 disparities");
    h->max_speckle_diff = v;
    return GIP_OK;
}
double gip_get_max_speckle_diff(gip_handle h) { return h ? h->max_speckle_diff : 0.0; }
int gip_set_pre_filter_size(gip_handle h, int v)
{
    GIP_SETTER(v >= 5 && v <= 255 && (v & 1), "setPreFilterSize: odd, 5..255");
    h->params.prefilter_size = v;
    return GIP_OK;
}
int gip_set_pre_filter_cap(gip_handle h, int v)
{
    GIP_SETTER(v >= 1 && v <= 63, "setPreFilterCap: 1..63");
    h->params.prefilter_cap = v;
    return GIP_OK;
}
int gip_set_uniqueness_ratio(gip_handle h, int v)
{
    GIP_SETTER(v >= 0, "setUniquenessRatio: >= 0");
    h->params.uniqueness_ratio = v;
    return GIP_OK;
}
int gip_set_disp12_max_diff(gip_handle h, int v) { H_CHECK(h); h->params.disp12_max_diff = v; return GIP_OK; }
#undef GIP_SETTER
int gip_get_params(gip_handle h, gip_bm_params* out) { H_CHECK(h); if (!out) return GIP_ERR_INVALID_ARG; *out = h->params; return GIP_OK; }
int gip_set_params(gip_handle h, const gip_bm_params* p)
{
    H_CHECK(h);
    if (!p) return GIP_ERR_INVALID_ARG;
    // the whole block at once: size-independent part of StereoBMImpl::compute's checks (the size-dependent one, block < min(w, h),
    // stays in computeDisparity)
    if (gip_validate_bm_params(p, 1 << 30, 1 << 30) != GIP_OK) return fail(h, GIP_ERR_INVALID_ARG, "setParams: invalid StereoBM parameter block");
    h->params = *p;
    return GIP_OK;
}

// ------------------------------------------------------------------------------------------------ imageCb schedule
int gip_process_pair(gip_handle h, const void* left_raw, const void* right_raw, int rows, int cols, size_t step, int type, const char* encoding,
                     uint32_t c)
{
    H_CHECK(h);
    if (!encoding || !encoding[0]) return fail(h, GIP_ERR_INVALID_ARG, "process_pair: empty encoding");
    int rc;
#define STEP(expr) do { rc = (expr); if (rc != GIP_OK) return rc; } while (0)
    STEP(gip_clean_senders(h));                                                                // StereoProcessor.cpp:166
    STEP(gip_upload_mat(h, GIP_SIDE_L | GIP_SRC_RAW, left_raw, rows, cols, step, type, encoding));    // :179
    STEP(gip_upload_mat(h, GIP_SIDE_R | GIP_SRC_RAW, right_raw, rows, cols, step, type, encoding));   // :180
    const bool disp_any = c & (GIP_TOPIC_DISPARITY | GIP_TOPIC_DISPARITY_VIS | GIP_TOPIC_POINTCLOUD);
    if ((c & (GIP_TOPIC_DEBAYER_MONO_LEFT | GIP_TOPIC_RECTIFY_MONO_LEFT)) || disp_any) STEP(gip_convert_raw_to_mono(h, GIP_SIDE_L));   // :183-186
    if ((c & (GIP_TOPIC_DEBAYER_MONO_RIGHT | GIP_TOPIC_RECTIFY_MONO_RIGHT)) || disp_any) STEP(gip_convert_raw_to_mono(h, GIP_SIDE_R)); // :192-195
    if (c & (GIP_TOPIC_DEBAYER_COLOR_LEFT | GIP_TOPIC_RECTIFY_COLOR_LEFT | GIP_TOPIC_POINTCLOUD)) STEP(gip_convert_raw_to_color(h, GIP_SIDE_L));  // :201-204
    if (c & (GIP_TOPIC_DEBAYER_COLOR_RIGHT | GIP_TOPIC_RECTIFY_COLOR_RIGHT)) STEP(gip_convert_raw_to_color(h, GIP_SIDE_R));                       // :210-213
    if ((c & GIP_TOPIC_RECTIFY_MONO_LEFT) || disp_any) STEP(gip_rectify_image(h, GIP_SIDE_L | GIP_SRC_MONO, GIP_SIDE_L | GIP_SRC_RECT_MONO, GIP_INTER_LINEAR));   // :220-223
    if ((c & GIP_TOPIC_RECTIFY_MONO_RIGHT) || disp_any) STEP(gip_rectify_image(h, GIP_SIDE_R | GIP_SRC_MONO, GIP_SIDE_R | GIP_SRC_RECT_MONO, GIP_INTER_LINEAR));  // :224-227
    if (c & (GIP_TOPIC_RECTIFY_COLOR_LEFT | GIP_TOPIC_POINTCLOUD)) STEP(gip_rectify_image(h, GIP_SIDE_L | GIP_SRC_COLOR, GIP_SIDE_L | GIP_SRC_RECT_COLOR, GIP_INTER_LINEAR));  // :239-242
    if (c & GIP_TOPIC_RECTIFY_COLOR_RIGHT) STEP(gip_rectify_image(h, GIP_SIDE_R | GIP_SRC_COLOR, GIP_SIDE_R | GIP_SRC_RECT_COLOR, GIP_INTER_LINEAR));                         // :243-246
    if (disp_any) {
        STEP(gip_compute_disparity(h, GIP_SIDE_L | GIP_SRC_RECT_MONO, GIP_SIDE_R | GIP_SRC_RECT_MONO, GIP_SIDE_L | GIP_SRC_DISPARITY));   // :261
        STEP(gip_filter_speckles(h, GIP_SIDE_L | GIP_SRC_DISPARITY));                                                                       // :262
    }
    if (c & GIP_TOPIC_DISPARITY_VIS) STEP(gip_compute_disparity_image(h, GIP_SIDE_L | GIP_SRC_DISPARITY, GIP_SIDE_L | GIP_SRC_DISPARITY_IMG));      // :273
    if (c & GIP_TOPIC_POINTCLOUD) STEP(gip_project_disparity_to_3d_points(h, GIP_SIDE_L | GIP_SRC_DISPARITY, GIP_SIDE_L | GIP_SRC_POINTS2));       // :280
#undef STEP
    return gip_wait_for_all_streams(h);                                                                                                      // :284
}

// ------------------------------------------------------------------------------------------------ batch pipeline
int gip_batch_set_maps_device(gip_handle h, int width, int height, const int16_t* map1_l, const uint16_t* map2_l, const int16_t* map1_r, const uint16_t* map2_r)
{
    H_CHECK(h);
    if (!map1_l || !map2_l || !map1_r || !map2_r || width <= 0 || height <= 0) return fail(h, GIP_ERR_INVALID_ARG, "batch_set_maps_device: bad arguments");
    if (h->maps_owned)
        for (int i = 0; i < 2; i++) {
            if (h->d_map1[i]) (void)hipFree(h->d_map1[i]);
            if (h->d_map2[i]) (void)hipFree(h->d_map2[i]);
        }
    h->maps_owned = false;
    h->d_map1[0] = (int16_t*)map1_l; h->d_map2[0] = (uint16_t*)map2_l;
    h->d_map1[1] = (int16_t*)map1_r; h->d_map2[1] = (uint16_t*)map2_r;
    h->map_w = width; h->map_h = height;
    return GIP_OK;
}

int gip_batch_run(gip_handle h, const gip_batch_desc* d, gip_stream stream)
{
    H_CHECK(h);
    if (!d || !d->left_raw || !d->right_raw || !d->disparity || d->width <= 0 || d->height <= 0 || d->batch <= 0)
        return fail(h, GIP_ERR_INVALID_ARG, "batch_run: bad descriptor");
    if (!h->model.initialised) return fail(h, GIP_ERR_NOT_INITIALISED, "batch_run: stereo model not initialised");
    if (!h->d_map1[0] || d->width != h->map_w || d->height != h->map_h) return fail(h, GIP_ERR_INVALID_ARG, "batch_run: image size differs from the rectification tables");
    H_HIP(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const int w = d->width, hh = d->height, B = d->batch;
    const size_t npx = (size_t)w * hh;
    const int enc = d->encoding;
    const int raw_cn = (enc == 1 || enc == 2) ? 3 : 1;
    const bool want_pc = (d->outputs & GIP_TOPIC_POINTCLOUD) && d->pointcloud;
    int rc = gip_validate_bm_params(&h->params, w, hh);
    if (rc != GIP_OK) return fail(h, rc, "batch_run: StereoBM parameter validation failed");
    gip_bm_params bp = h->params;
    bp.speckle_window_size = 0;  // the node-level speckle filter below is the one the reference runs
    // gip_batch_desc::phase: FRONT enqueues everything up to and including the matcher (which, on a matcher stream of its own, is NOT joined
    // back), BACK waits for that matcher and enqueues the rest.  The pointer arithmetic below runs in both; `front` / `back` gate the launches.
    const int phase = d->phase;
    if (phase < GIP_BATCH_WHOLE || phase > GIP_BATCH_BACK) return fail(h, GIP_ERR_INVALID_ARG, "batch_run: bad phase");
    if (phase == GIP_BATCH_BACK && !h->front_pending) return fail(h, GIP_ERR_INVALID_ARG, "batch_run: BACK phase without a FRONT phase before it");
    if (phase != GIP_BATCH_BACK && h->front_pending) return fail(h, GIP_ERR_INVALID_ARG, "batch_run: the FRONT phase of the previous batch has not been completed by its BACK phase");
    const bool front = phase != GIP_BATCH_BACK, back = phase != GIP_BATCH_FRONT;
#define B_STEP(expr, what) do { rc = (expr); if (rc != GIP_OK) return propagate(h, rc, what); } while (0)
    if (h->timing_runs >= 4096) h->timing_runs = 0;  // ring: the caller is expected to read/reset well before this
    while ((int)h->tev.size() < 5 * (h->timing_runs + 1)) {
        hipEvent_t e;
        H_HIP(h, hipEventCreate(&e));
        h->tev.push_back(e);
    }
    hipEvent_t* ev = &h->tev[5 * h->timing_runs];
    // FRONT and BACK may run on different streams (all FRONT phases in order on one, all BACK phases on another: neither chain then waits for the
    // other): the buffers of this handle are free for the next FRONT only when the previous BACK is through with them
    if (phase == GIP_BATCH_FRONT && h->back_done_valid) H_HIP(h, hipStreamWaitEvent(st, h->back_done, 0));
    if (front) H_HIP(h, hipEventRecord(ev[2], st));
    // stage 1: colour (StereoProcessor.cpp:183-217)
    const uint8_t* mono[2] = {d->left_raw, d->right_raw};
    const uint8_t* color_l = nullptr;
    if (enc != 0) {
        for (int i = 0; i < 2; i++) {
            B_STEP(ensure_dev(h, h->b_mono[i], npx * B), "batch buffers");
            const uint8_t* src = i == 0 ? d->left_raw : d->right_raw;
            if (!front) { mono[i] = (const uint8_t*)h->b_mono[i].p; continue; }
            if (enc == 1 || enc == 2) B_STEP(gipk_cvt_bgr_to_gray(src, (size_t)w * 3, (uint8_t*)h->b_mono[i].p, w, w, hh, enc == 2, B, npx * 3, npx, st), "bgr->gray");
            else B_STEP(gipk_cvt_bayer_to_gray(src, w, (uint8_t*)h->b_mono[i].p, w, w, hh, enc - 3, B, npx, npx, st), "bayer->gray");
            mono[i] = (const uint8_t*)h->b_mono[i].p;
        }
        if (want_pc) {
            if (enc == 1) color_l = d->left_raw;
            else {
                B_STEP(ensure_dev(h, h->b_color, npx * 3 * B), "batch buffers");
                if (!front) {}
                else if (enc == 2) B_STEP(gipk_cvt_swap_rb(d->left_raw, (size_t)w * 3, (uint8_t*)h->b_color.p, (size_t)w * 3, w, hh, B, npx * 3, npx * 3, st), "rgb->bgr");
                else B_STEP(gipk_cvt_bayer_to_bgr(d->left_raw, w, (uint8_t*)h->b_color.p, (size_t)w * 3, w, hh, enc - 3, B, npx, npx * 3, st), "bayer->bgr");
                color_l = (const uint8_t*)h->b_color.p;
            }
        }
    }
    (void)raw_cn;
    // stage 2: rectify (StereoProcessor.cpp:220-256) + the matcher's prefilter.  gip_batch_desc::fused_front_end selects the
    // one-pass form raw -> prefiltered plane (k_rectify_xsobel; the rectified plane is then written only where something else reads
    // it: the left one, as the cloud's colour for mono input); NORMALIZED_RESPONSE / undistorted cameras always take two passes.
    // rectified mono planes: rows padded to 64 bytes so the prefilter can work on aligned dwords
    const size_t rp = gip_align_up((size_t)w, 64), rplane = rp * hh;
    B_STEP(ensure_dev(h, h->bm_scratch, gipk_stereobm_scratch_bytes(w, hh, B, &bp)), "batch buffers");
    uint8_t* pf[2];
    size_t pfp = 0, pfplane = 0;
    gip_stereobm_planes(h->bm_scratch.p, w, hh, B, &pf[0], &pf[1], &pfp, &pfplane);
    const bool need_rect_l = want_pc && !color_l;     // mono input: the rectified left image colours the cloud
    bool fused[2] = {false, false};
    for (int i = 0; i < 2; i++) {
        const GipPinhole& cam = i == 0 ? h->model.left : h->model.right;
        const bool need_rect = i == 0 && need_rect_l;
        fused[i] = cam.distorted && bp.prefilter_type == GIP_PREFILTER_XSOBEL && d->fused_front_end;
        if (need_rect || !fused[i]) B_STEP(ensure_dev(h, h->b_rect[i], rplane * B), "batch buffers");
        if (!front) continue;
        if (fused[i])
            B_STEP(gipk_rectify_xsobel(mono[i], w, w, hh, pf[i], pfp, need_rect ? (uint8_t*)h->b_rect[i].p : nullptr, rp, w, hh, h->d_map1[i], h->d_map2[i], bp.prefilter_cap, B, npx,
                                       pfplane, rplane, st), "rectify + prefilter");
        else if (cam.distorted)
            B_STEP(gipk_remap_u8(mono[i], w, w, hh, (uint8_t*)h->b_rect[i].p, rp, w, hh, 1, h->d_map1[i], h->d_map2[i], B, npx, rplane, st), "rectify");
        else
            H_HIP(h, hipMemcpy2DAsync(h->b_rect[i].p, rp, mono[i], w, w, (size_t)hh * B, hipMemcpyDeviceToDevice, st));
        if (!fused[i])
            B_STEP(gipk_prefilter((const uint8_t*)h->b_rect[i].p, rp, pf[i], pfp, w, hh, bp.prefilter_type, bp.prefilter_size, bp.prefilter_cap, B, rplane, pfplane, st), "prefilter");
    }
    const uint8_t* rect_color = nullptr;
    int rect_color_cn = 3;
    if (want_pc) {
        if (color_l) {
            B_STEP(ensure_dev(h, h->b_rect_color, npx * 3 * B), "batch buffers");
            if (!front) {}
            else if (h->model.left.distorted)
                B_STEP(gipk_remap_u8(color_l, (size_t)w * 3, w, hh, (uint8_t*)h->b_rect_color.p, (size_t)w * 3, w, hh, 3, h->d_map1[0], h->d_map2[0], B, npx * 3, npx * 3, st), "rectify colour");
            else
                H_HIP(h, hipMemcpyAsync(h->b_rect_color.p, color_l, npx * 3 * B, hipMemcpyDeviceToDevice, st));
            rect_color = (const uint8_t*)h->b_rect_color.p;
        } else {
            // mono8 input: GRAY2BGR then remap == remap then GRAY2BGR (every channel sees the same taps and weights), and the
            // replication itself is folded into the cloud kernel (colour channels = 1)
            rect_color = (const uint8_t*)h->b_rect[0].p;
            rect_color_cn = 1;
        }
    }
    // stage 3: disparity + speckle (StereoProcessor.cpp:259-268)
    // launch context of the matcher: timing events on the stream it runs on, and (gip_batch_desc::matcher_stream) a stream of
    // its own fenced by a fork / join event pair
    BmLaunchCtx mctx;
    mctx.ev0 = ev[0];
    mctx.ev1 = ev[1];
    mctx.co_scheduled = d->batches_in_flight > 1 || (d->matcher_stream && (hipStream_t)d->matcher_stream != st);
    mctx.defer_join = phase == GIP_BATCH_FRONT;
    if (d->matcher_stream && (hipStream_t)d->matcher_stream != st) {
        if (!h->match_fork) {
            H_HIP(h, hipEventCreateWithFlags(&h->match_fork, hipEventDisableTiming));
            H_HIP(h, hipEventCreateWithFlags(&h->match_join, hipEventDisableTiming));
        }
        mctx.matcher_stream = (hipStream_t)d->matcher_stream;
        mctx.fork = h->match_fork;
        mctx.join = h->match_join;
    }
    // both prefiltered planes are in the scratch already (null inputs)
    if (phase == GIP_BATCH_BACK && h->front_join) H_HIP(h, hipStreamWaitEvent(st, h->front_join, 0));
    if (front) h->front_cost_u16 = 1;
    rc = gip_stereobm_phase_impl(nullptr, nullptr, pfp, w, hh, &bp, d->disparity, (size_t)w * 2, h->bm_scratch.p, h->bm_scratch.cap, B, pfplane, npx * 2, st, &mctx, phase,
                                 &h->front_cost_u16);
    if (rc != GIP_OK) return propagate(h, rc, "stereobm");
    if (phase == GIP_BATCH_FRONT) {
        h->front_pending = true;
        h->front_join = mctx.pending_join;   // null when the matcher ran on `st` itself
        return GIP_OK;
    }
    h->front_pending = false;
    (void)back;
    H_HIP(h, hipEventRecord(ev[4], st));   // end of the "remap + SAD" span: colour, rectify, prefilter, matcher, validateDisparity
    if (h->max_speckle_size > 0) {
        B_STEP(ensure_dev(h, h->spk_scratch, gipk_speckle_scratch_bytes(w, hh, B)), "batch buffers");
        B_STEP(gipk_filter_speckles(d->disparity, (size_t)w * 2, w, hh, (bp.min_disparity - 1) * 16, h->max_speckle_size, (int)(h->max_speckle_diff * 16.0), h->spk_scratch.p,
                                    h->spk_scratch.cap, B, npx * 2, st), "speckle");
    }
    // stage 5: float disparity (GPUStereoProcessor.cpp:320) + cloud (StereoProcessor.cpp:278-282, GpuSenderPc2.cpp) in one pass
    if (d->disparity_f32 || want_pc) {
        const void* tab = nullptr;
#ifndef GIP_NO_W_TABLE
        if (want_pc) {
            const double key[6] = {h->model.Q[11], h->model.Q[14], h->model.Q[15], h->model.left.cx() - h->model.right.cx(), (double)bp.min_disparity, (double)bp.num_disparities};
            if (!h->w_valid || std::memcmp(key, h->w_key, sizeof key) != 0) {
                B_STEP(ensure_dev(h, h->w_tab, gip_w_table_bytes(bp.num_disparities)), "batch buffers");
                B_STEP(gip_w_table_build(h->w_tab.p, h->model.Q, h->model.left.cx(), h->model.right.cx(), bp.min_disparity, bp.num_disparities, st), "disparity table");
                std::memcpy(h->w_key, key, sizeof key);
                h->w_valid = true;
            }
            tab = h->w_tab.p;
        }
#endif
        B_STEP(gip_disparity_outputs_impl(d->disparity, (size_t)w * 2, rect_color, rect_color_cn == 1 ? rp : (size_t)w * 3, rect_color_cn, d->disparity_f32,
                                          want_pc ? d->pointcloud : nullptr, w, hh, 0, h->model.Q, h->model.left.cx(), h->model.right.cx(), bp.min_disparity, B, npx * 2,
                                          rect_color_cn == 1 ? rplane : npx * 3, st, tab, bp.num_disparities, true), "disparity outputs");
    }
    H_HIP(h, hipEventRecord(ev[3], st));
    if (phase == GIP_BATCH_BACK) {
        if (!h->back_done) H_HIP(h, hipEventCreateWithFlags(&h->back_done, hipEventDisableTiming));
        H_HIP(h, hipEventRecord(h->back_done, st));
        h->back_done_valid = true;
    }
    h->timing_runs++;
#undef B_STEP
    return GIP_OK;
}

int gip_batch_last_kernel_ms(gip_handle h, float* bm_ms, float* total_ms)
{
    H_CHECK(h);
    if (h->timing_runs <= 0) return fail(h, GIP_ERR_INVALID_ARG, "no batch has run");
    hipEvent_t* ev = &h->tev[5 * (h->timing_runs - 1)];
    if (bm_ms) H_HIP(h, hipEventElapsedTime(bm_ms, ev[0], ev[1]));
    if (total_ms) H_HIP(h, hipEventElapsedTime(total_ms, ev[2], ev[3]));
    return GIP_OK;
}

int gip_batch_timing_reset(gip_handle h)
{
    H_CHECK(h);
    h->timing_runs = 0;
    return GIP_OK;
}

int gip_batch_timing_read(gip_handle h, double* bm_ms_sum, double* total_ms_sum, int* runs) { return gip_batch_timing_read_spans(h, bm_ms_sum, total_ms_sum, nullptr, runs); }

int gip_batch_timing_read_spans(gip_handle h, double* bm_ms_sum, double* total_ms_sum, double* remap_sad_ms_sum, int* runs)
{
    H_CHECK(h);
    double a = 0, b = 0, c = 0;
    for (int i = 0; i < h->timing_runs; i++) {
        float x = 0, y = 0, z = 0;
        H_HIP(h, hipEventElapsedTime(&x, h->tev[5 * i], h->tev[5 * i + 1]));
        H_HIP(h, hipEventElapsedTime(&y, h->tev[5 * i + 2], h->tev[5 * i + 3]));
        H_HIP(h, hipEventElapsedTime(&z, h->tev[5 * i + 2], h->tev[5 * i + 4]));
        a += x;
        b += y;
        c += z;
    }
    if (bm_ms_sum) *bm_ms_sum = a;
    if (total_ms_sum) *total_ms_sum = b;
    if (remap_sad_ms_sum) *remap_sad_ms_sum = c;
    if (runs) *runs = h->timing_runs;
    return GIP_OK;
}

}  // extern "C"
