// gip_streams.hip -- streams that really run side by side (gip_concurrent_streams_*).
//
// Where the reference keeps one cv::cuda::Stream per camera side and trusts the driver to overlap them (GPUStereoProcessor.cpp:190-200
// getStream), the throughput pipelines here depend on FOUR streams making progress at the same time: three batches' small kernels next
// to the shared matcher stream.  The HIP runtime maps the streams of a process onto a few hardware queues
// (GPU_MAX_HW_QUEUES, 4 unless the user raises it) by a policy of its own, and the command processor serves hardware queues through four
// pipes.  Two streams on one queue run strictly one after the other, two queues on one pipe take turns in units of about a millisecond
// (measured: tools/probe/queue_conflict.hip, profiles/r03_queue_conflict.txt): a pipeline whose matcher stream shares a queue with one of
// its three batch streams loses 12-25 % (49.6 k against 55-57 k pairs/s, same binary, only the order the streams were created in
// changed).  The mapping is neither documented nor queryable, so it is MEASURED: candidates are created one at a time and kept only if a
// one-workgroup kernel on the candidate finishes while a chip-filling kernel runs on every stream kept so far.
#include <hip/hip_runtime.h>

#include <chrono>
#include <vector>

#include "gip_internal.h"
#include "gpuimageproc_hip.h"

namespace {

// every workgroup spins for `ticks` of the 100 MHz constant clock.  Launched with 60 KB of LDS per workgroup, two of them fit on a CU:
// the chip's LDS is taken but wave slots stay free, so a kernel without LDS on a stream that is NOT held up by the dispatcher starts at once.
__global__ void k_probe_spin(unsigned long long ticks, unsigned* sink)
{
    const unsigned long long t0 = wall_clock64();
    unsigned x = threadIdx.x;
    while (wall_clock64() - t0 < ticks) x = x * 1664525u + 1013904223u;
    if (x == 0xFFFFFFFFu && sink) sink[0] = x;    // never true for the values reached; keeps the loop
}
__global__ void k_probe_touch(unsigned* out)
{
    if (threadIdx.x == 0) out[0] = 1;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Probe {
    int n_cu = 256;
    unsigned* d = nullptr;
    // true when a kernel on `b` is held up by a chip-filling kernel on `a`
    int conflict(hipStream_t a, hipStream_t b, bool* out)
    {
        // 6 rounds of resident workgroups x 500 us = 3 ms: long against the millisecond in which two queues of one pipe take turns
        const unsigned long long ticks = 50000;
        const size_t lds = 60 * 1024;
        hipLaunchKernelGGL(k_probe_spin, dim3(n_cu * 2 * 6), dim3(64), lds, a, ticks, d);
        GIP_LAUNCH_CHECK();
        const double w0 = now_s();
        while (now_s() - w0 < 200e-6) {}          // let the dispatcher pick the big kernel up before the small one is queued
        const double t0 = now_s();
        hipLaunchKernelGGL(k_probe_touch, dim3(1), dim3(64), 0, b, d + 64);
        GIP_LAUNCH_CHECK();
        GIP_HIP_CHECK(hipStreamSynchronize(b));
        const double tb = now_s() - t0;
        GIP_HIP_CHECK(hipStreamSynchronize(a));
        const double ta = now_s() - t0;
        *out = tb > 0.1 * ta;                      // concurrent: tens of microseconds against 3 ms
        return GIP_OK;
    }
};

}  // namespace

int gip_concurrent_streams_create(int device, int n, gip_stream* streams, int* n_independent)
{
    if (n <= 0 || n > 16 || !streams) return GIP_ERR_INVALID_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return GIP_ERR_NO_DEVICE;
    GIP_HIP_CHECK(hipSetDevice(device));
    Probe pr;
    hipDeviceProp_t prop;
    GIP_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    pr.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    GIP_HIP_CHECK(hipMalloc((void**)&pr.d, 4096));
    if (hipMemset(pr.d, 0, 4096) != hipSuccess) {
        (void)hipFree(pr.d);
        return GIP_ERR_HIP;
    }
    std::vector<hipStream_t> kept, spare;
    int rc = GIP_OK;
    auto make = [&](hipStream_t* s) -> int {
        GIP_HIP_CHECK(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
        hipLaunchKernelGGL(k_probe_touch, dim3(1), dim3(64), 0, *s, pr.d + 64);    // the first launch sets the stream's queue up
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(*s) != hipSuccess) {
            (void)hipStreamDestroy(*s);
            *s = nullptr;
            return GIP_ERR_HIP;
        }
        return GIP_OK;
    };
    // rejected candidates stay alive until the end: the runtime hands a new stream the least loaded queue, so they steer the next
    // candidate somewhere else
    for (int tries = 0; (int)kept.size() < n && tries < 4 * n + 8 && rc == GIP_OK; tries++) {
        hipStream_t c = nullptr;
        rc = make(&c);
        if (rc != GIP_OK) break;
        bool clash = false;
        for (size_t i = 0; i < kept.size() && !clash && rc == GIP_OK; i++) rc = pr.conflict(kept[i], c, &clash);
        (clash || rc != GIP_OK ? spare : kept).push_back(c);     // (on an error the candidate is released with the spares below)
    }
    const int independent = (int)kept.size();
    while (rc == GIP_OK && (int)kept.size() < n) {       // not enough independent queues (n above GPU_MAX_HW_QUEUES): fill up with what there is
        if (spare.empty()) {
            hipStream_t c = nullptr;
            rc = make(&c);
            if (rc == GIP_OK) kept.push_back(c);
        } else {
            kept.push_back(spare.back());
            spare.pop_back();
        }
    }
    for (hipStream_t s : spare) (void)hipStreamDestroy(s);
    (void)hipFree(pr.d);
    if (rc != GIP_OK) {
        for (hipStream_t s : kept) (void)hipStreamDestroy(s);
        return rc;
    }
    for (int i = 0; i < n; i++) streams[i] = (gip_stream)kept[i];
    if (n_independent) *n_independent = independent < n ? independent : n;
    return GIP_OK;
}

int gip_concurrent_streams_destroy(int device, int n, gip_stream* streams)
{
    if (n < 0 || (n > 0 && !streams)) return GIP_ERR_INVALID_ARG;
    if (hipSetDevice(device) != hipSuccess) return GIP_ERR_NO_DEVICE;
    for (int i = 0; i < n; i++)
        if (streams[i]) {
            (void)hipStreamSynchronize((hipStream_t)streams[i]);
            (void)hipStreamDestroy((hipStream_t)streams[i]);
            streams[i] = nullptr;
        }
    return GIP_OK;
}
