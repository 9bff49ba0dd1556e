// gip_bench.cpp -- the throughput measurement of bench.py from a plain C++ host: no Python, no torch, only the C-ABI (gpuimageproc_hip.h) and
// the HIP runtime for device memory and streams.  What a C++ integrator (the reference's nodelet is C++: src/StereoProcessor.cpp) links against.
//
//   gip_bench [--width 1242] [--height 375] [--ndisp 128] [--block 9] [--batch 256] [--depth 3] [--steps 200] [--warmup 10]
//             [--streams 2]            two streams in all: FRONT(k+1) / BACK(k) phases of gip_batch_run in order on one, the matchers on the other
//   stream-mapping experiments (profiles/r03_stream_mapping.txt):
//             [--plain-streams 1]      hipStreamCreate in program order instead of gip_concurrent_streams_create
//             [--idle-before N] [--idle-between N] [--matcher-last 1] [--matcher-priority P] [--pipe-priority P]   (with --plain-streams 1)
//
// `depth` processor handles, each with its own stream and output buffers, take the steps in turn; the SAD matchers of all of them go to ONE
// shared stream (gip_batch_desc::matcher_stream), exactly as bench.py does.  The streams come as one set measured to dispatch side by side
// (gip_concurrent_streams_create).  Inputs are synthetic (textured noise with planted disparities,
// resident in HBM before the timed region); the calibration is the reference fixture's left camera (test/stereobm/test_data/left.yaml) scaled
// to the image size.  Prints one JSON line.  Results are not checked here -- parity lives in tests/ (this binary never links the oracle).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "gpuimageproc_hip.h"

#define CK_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s failed: %s\n", #expr, hipGetErrorString(e_)); return 2; } \
    } while (0)
#define CK_GIP(h, expr)                                                                                \
    do {                                                                                               \
        int rc_ = (expr);                                                                              \
        if (rc_ != GIP_OK) { std::fprintf(stderr, "%s: %s (%s)\n", #expr, gip_status_string(rc_), (h) ? gip_last_error(h) : ""); return 2; } \
    } while (0)

// the LEFT camera of the reference fixture (test/stereobm/test_data/left.yaml, 752x480) scaled to w x h, used for both eyes (the right one with
// the fixture's Tx * fx): real lens distortion and a rotated rectification, which is all a throughput run needs from a calibration
static void scaled_camera(int w, int h, bool right, gip_camera_info* ci)
{
    std::memset(ci, 0, sizeof *ci);
    const double sx = w / 752.0, sy = h / 480.0;
    const double K[9] = {463.241134, 0, 385.760092, 0, 463.072782, 221.039882, 0, 0, 1};
    const double D[5] = {-0.374450, 0.109252, 0.000282, -0.001548, 0.0};
    const double Rm[9] = {0.997310, 0.005745, 0.073072, -0.006100, 0.999971, 0.004637, -0.073043, -0.005070, 0.997316};
    const double P[12] = {441.238411, 0, 322.811100, 0, 0, 441.238411, 230.623768, 0, 0, 0, 1, 0};
    ci->width = w; ci->height = h;
    for (int i = 0; i < 9; i++) { ci->K[i] = K[i]; ci->R[i] = Rm[i]; }
    for (int i = 0; i < 12; i++) ci->P[i] = P[i];
    ci->K[0] *= sx; ci->K[2] *= sx; ci->K[4] *= sy; ci->K[5] *= sy;
    ci->P[0] *= sx; ci->P[2] *= sx; ci->P[5] *= sy; ci->P[6] *= sy;
    for (int i = 0; i < 5; i++) ci->D[i] = D[i];
    ci->nD = 5;
    if (right) ci->P[3] = -44.133133 * sx;
}

int main(int argc, char** argv)
{
    int w = 1242, h = 375, nd = 128, block = 9, B = 256, depth = 3, steps = 200, warmup = 10, idle_before = 0, idle_between = 0, matcher_last = 0, mprio = 0, pprio = 0, streams = 0, plain = 0;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string k = argv[i];
        const int v = std::atoi(argv[i + 1]);
        if (k == "--width") w = v; else if (k == "--height") h = v; else if (k == "--ndisp") nd = v; else if (k == "--block") block = v;
        else if (k == "--batch") B = v; else if (k == "--depth") depth = v; else if (k == "--steps") steps = v; else if (k == "--warmup") warmup = v;
        else if (k == "--idle-before") idle_before = v; else if (k == "--idle-between") idle_between = v; else if (k == "--matcher-last") matcher_last = v;
        else if (k == "--streams") streams = v; else if (k == "--plain-streams") plain = v; else if (k == "--matcher-priority") mprio = v; else if (k == "--pipe-priority") pprio = v;
        else { std::fprintf(stderr, "unknown option %s\n", k.c_str()); return 1; }
    }
    if (gip_device_count() <= 0) { std::fprintf(stderr, "gip_bench needs a GPU (there is no CPU fallback)\n"); return 3; }
    const size_t npx = (size_t)w * h;
    // synthetic pairs: smooth noise texture, right(x) = left(x + d) with d changing every 48 rows; four distinct scenes repeated over the batch
    std::vector<uint8_t> L(npx * B), R(npx * B);
    std::mt19937 rng(12345);
    for (int s = 0; s < 4; s++) {
        std::vector<uint8_t> tex((size_t)h * (w + nd));
        for (auto& v : tex) v = (uint8_t)(rng() & 0xFF);
        for (int y = 0; y < h; y++)                                  // 1-2-1 horizontal smoothing: texture with some spatial correlation
            for (int x = 1; x + 1 < w + nd; x++) tex[(size_t)y * (w + nd) + x] = (uint8_t)((tex[(size_t)y * (w + nd) + x - 1] + 2 * tex[(size_t)y * (w + nd) + x] + tex[(size_t)y * (w + nd) + x + 1]) >> 2);
        for (int y = 0; y < h; y++) {
            const int d = 4 + (int)((y / 48 * 37 + s * 11) % (nd - 12));
            for (int x = 0; x < w; x++) {
                R[(size_t)s * npx + (size_t)y * w + x] = tex[(size_t)y * (w + nd) + x + nd];
                L[(size_t)s * npx + (size_t)y * w + x] = tex[(size_t)y * (w + nd) + x + nd - d];
            }
        }
    }
    for (int b = 4; b < B; b++) {
        std::memcpy(&L[(size_t)b * npx], &L[(size_t)(b % 4) * npx], npx);
        std::memcpy(&R[(size_t)b * npx], &R[(size_t)(b % 4) * npx], npx);
    }
    gip_camera_info cl, cr;
    scaled_camera(w, h, false, &cl);
    scaled_camera(w, h, true, &cr);
    gip_bm_params p{};
    p.prefilter_type = GIP_PREFILTER_XSOBEL; p.prefilter_size = 9; p.prefilter_cap = 31; p.block_size = block; p.min_disparity = 0; p.num_disparities = nd;
    p.texture_threshold = 10; p.uniqueness_ratio = 15; p.disp12_max_diff = 1;

    struct Pipe { gip_handle h = nullptr; hipStream_t st = nullptr; uint8_t *l = nullptr, *r = nullptr, *pc = nullptr; int16_t* d16 = nullptr; float* f32 = nullptr; };
    std::vector<Pipe> pipes(depth);
    hipStream_t matcher = nullptr;
    std::vector<hipStream_t> idle;                                   // --idle-*: streams that are created and never used (hardware-queue mapping experiments)
    auto add_idle = [&](int n) { for (int i = 0; i < n; i++) { hipStream_t s; if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess) idle.push_back(s); } };
    add_idle(idle_before);
    // the pipeline's streams: batch streams + the matcher stream as ONE set of mutually concurrent streams (gip_concurrent_streams_create);
    // --plain-streams 1 creates them with hipStreamCreate in program order instead (the experiments of DESIGN.md section 4)
    std::vector<gip_stream> cs;
    int n_indep = 0;
    if (!plain) {
        cs.resize((streams == 2 ? 1 : depth) + (depth > 1 ? 1 : 0));
        const int rc_cs = gip_concurrent_streams_create(0, (int)cs.size(), cs.data(), &n_indep);
        if (rc_cs != GIP_OK) { std::fprintf(stderr, "gip_concurrent_streams_create: %s\n", gip_status_string(rc_cs)); return 2; }
        if (depth > 1) matcher = (hipStream_t)cs.back();
    }
    if (plain && depth > 1 && !matcher_last) CK_HIP(hipStreamCreateWithPriority(&matcher, hipStreamNonBlocking, mprio));
    for (Pipe& q : pipes) {
        CK_GIP(nullptr, gip_create(0, &q.h));
        CK_GIP(q.h, gip_init_stereo_model(q.h, &cl, &cr));
        CK_GIP(q.h, gip_set_params(q.h, &p));
        CK_GIP(q.h, gip_set_max_speckle_size(q.h, 800));
        CK_GIP(q.h, gip_set_max_speckle_diff(q.h, 5.0));
        if (streams == 2 && &q != &pipes[0]) q.st = pipes[0].st;      // --streams 2: ONE stream for everything but the matcher (FRONT / BACK phases)
        else if (!plain) q.st = (hipStream_t)cs[&q - &pipes[0]];
        else CK_HIP(hipStreamCreateWithPriority(&q.st, hipStreamNonBlocking, pprio));
        add_idle(idle_between);
        CK_HIP(hipMalloc((void**)&q.l, npx * B));
        CK_HIP(hipMalloc((void**)&q.r, npx * B));
        CK_HIP(hipMalloc((void**)&q.d16, npx * B * 2));
        CK_HIP(hipMalloc((void**)&q.f32, npx * B * 4));
        CK_HIP(hipMalloc((void**)&q.pc, npx * B * 32));
        CK_HIP(hipMemcpy(q.l, L.data(), npx * B, hipMemcpyHostToDevice));
        CK_HIP(hipMemcpy(q.r, R.data(), npx * B, hipMemcpyHostToDevice));
    }
    if (plain && depth > 1 && matcher_last) CK_HIP(hipStreamCreateWithPriority(&matcher, hipStreamNonBlocking, mprio));
    auto step = [&](int i, int phase = GIP_BATCH_WHOLE) -> int {
        Pipe& q = pipes[i % depth];
        gip_batch_desc d{};
        d.phase = phase;
        d.width = w; d.height = h; d.batch = B; d.encoding = 0;
        d.outputs = GIP_TOPIC_DISPARITY | GIP_TOPIC_POINTCLOUD;
        d.left_raw = q.l; d.right_raw = q.r; d.disparity = q.d16; d.disparity_f32 = q.f32; d.pointcloud = q.pc;
        d.matcher_stream = matcher;
        d.batches_in_flight = depth;
        CK_GIP(q.h, gip_batch_run(q.h, &d, q.st));
        return 0;
    };
    // --streams 2: software pipeline on two streams -- FRONT(i + 1) is enqueued before BACK(i), so the pipe stream always holds the next
    // batch's front end ahead of the wait for the current matcher
    auto run_steps = [&](int first, int n) -> int {
        if (streams != 2) {
            for (int i = first; i < first + n; i++)
                if (step(i)) return 2;
            return 0;
        }
        if (step(first, GIP_BATCH_FRONT)) return 2;
        for (int i = first; i < first + n; i++) {
            if (i + 1 < first + n && step(i + 1, GIP_BATCH_FRONT)) return 2;
            if (step(i, GIP_BATCH_BACK)) return 2;
        }
        return 0;
    };
    if (run_steps(0, warmup)) return 2;
    CK_HIP(hipDeviceSynchronize());
    for (Pipe& q : pipes) CK_GIP(q.h, gip_batch_timing_reset(q.h));
    const auto t0 = std::chrono::steady_clock::now();
    if (run_steps(warmup, steps)) return 2;
    CK_HIP(hipDeviceSynchronize());
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    double bm = 0, tot = 0;
    int runs = 0;
    for (Pipe& q : pipes) {
        double a = 0, b = 0;
        int n = 0;
        CK_GIP(q.h, gip_batch_timing_read(q.h, &a, &b, &n));
        bm += a; tot += b; runs += n;
    }
    // a glance at the result (not a parity check): fraction of valid disparities of the first image
    std::vector<int16_t> d0(npx);
    CK_HIP(hipMemcpy(d0.data(), pipes[(warmup + steps - 1) % depth].d16, npx * 2, hipMemcpyDeviceToHost));
    size_t valid = 0;
    for (int16_t v : d0) valid += v != -16;
    std::printf("{\"tool\": \"gip_bench (C++ host, C-ABI only)\", \"workload\": \"%dx%d mono8, ndisp=%d block=%d, full chain rectify->StereoBM->speckle(800,5)->f32 disparity+PointCloud2\", "
                "\"value\": %.2f, \"unit\": \"stereo pairs/s\", \"steps\": %d, \"warmup\": %d, \"pairs_per_step\": %d, \"batches_in_flight\": %d, \"independent_streams\": %d, \"ms_per_step\": %.4f, "
                "\"matcher_ms_per_launch\": %.4f, \"chain_gpu_ms_per_step\": %.4f, \"valid_fraction_first_image\": %.3f}\n",
                w, h, nd, block, (double)B * steps / dt, steps, warmup, B, depth, n_indep, 1e3 * dt / steps, runs ? bm / runs : 0.0, runs ? tot / runs : 0.0, (double)valid / npx);
    for (Pipe& q : pipes) {
        (void)hipFree(q.l); (void)hipFree(q.r); (void)hipFree(q.d16); (void)hipFree(q.f32); (void)hipFree(q.pc);
        if (plain && (streams != 2 || &q == &pipes[0])) (void)hipStreamDestroy(q.st);
        (void)gip_destroy(q.h);
    }
    if (plain && matcher) (void)hipStreamDestroy(matcher);
    if (!plain) (void)gip_concurrent_streams_destroy(0, (int)cs.size(), cs.data());
    for (hipStream_t s : idle) (void)hipStreamDestroy(s);
    return 0;
}
