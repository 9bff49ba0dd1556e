// queue_conflict.hip -- which HIP streams of a process dispatch concurrently on MI355X?
//
// The HIP runtime multiplexes the streams of a process onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4), and the command
// processor serves hardware queues through a handful of pipes.  A dispatch whose grid does not fit on the chip keeps its queue (and
// whatever else shares the dispatcher) busy until its last workgroup has been issued, so a small kernel on a stream that shares with it
// starts only after the big one -- the difference between 50 k and 57 k pairs/s in the throughput pipeline.
//
// The probe creates N streams one after the other and measures, for every ordered pair (i, j), how long a one-workgroup kernel on stream
// j takes to complete while a chip-filling kernel of several milliseconds runs on stream i.  Output: the N x N matrix of those
// latencies in microseconds ("blocked" entries are of the order of the big kernel's duration).
//   build + run (on the GPU box):  hipcc --offload-arch=gfx950 -O2 -o /tmp/queue_conflict tools/probe/queue_conflict.hip && /tmp/queue_conflict [N] [idle-between]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// every workgroup spins for `ticks` of the 100 MHz constant clock; 60 KB of LDS per workgroup keep two of them on a CU (wave slots stay
// free for the small kernel), and the grid is ten times that resident capacity
__global__ void k_block(unsigned long long ticks, unsigned* sink)
{
    const unsigned long long t0 = wall_clock64();
    unsigned x = threadIdx.x;
    while (wall_clock64() - t0 < ticks) x = x * 1664525u + 1013904223u;
    if (x == 0xFFFFFFFFu && sink) sink[0] = x;
}
__global__ void k_small(unsigned* out) { if (threadIdx.x == 0) out[0] += 1; }

int main(int argc, char** argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 8;
    const int with_null = argc > 2 ? std::atoi(argv[2]) : 0;       // 1: touch the null stream first (hipMemcpy), as most programs do
    unsigned* d = nullptr;
    CK(hipMalloc((void**)&d, 4096));
    CK(hipMemset(d, 0, 4096));
    if (with_null) { unsigned v = 0; CK(hipMemcpy(&v, d, 4, hipMemcpyDeviceToHost)); }
    std::vector<hipStream_t> st(n);
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // warm every stream (first launches create queues / load code objects)
    for (auto& s : st) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, d); CK(hipStreamSynchronize(s)); }
    const size_t lds = 60 * 1024;
    hipLaunchKernelGGL(k_block, dim3(256), dim3(64), lds, st[0], 100ull, d);
    CK(hipDeviceSynchronize());
    // big kernel: 256 CUs x 2 resident workgroups x 10 rounds of 500 us = 5 ms
    const int grid = 256 * 2 * 10;
    const unsigned long long ticks = 50000;   // 500 us at 100 MHz
    std::printf("# rows: stream of the chip-filling kernel; columns: stream of the one-workgroup kernel; entries: its completion latency in us\n");
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) {
            if (i == j) { std::printf("%7s", "-"); continue; }
            hipLaunchKernelGGL(k_block, dim3(grid), dim3(64), lds, st[i], ticks, d);
            // give the dispatcher time to start the big kernel before the small one is queued
            const auto w0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < 300e-6) {}
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st[j], d + 16);
            CK(hipStreamSynchronize(st[j]));
            const double us = 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            CK(hipStreamSynchronize(st[i]));
            std::printf("%7.0f", us);
        }
        std::printf("\n");
    }
    for (auto& s : st) (void)hipStreamDestroy(s);
    (void)hipFree(d);
    return 0;
}
