// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths our kernels use (the MI355X guide calibrates only
// 16-byte-per-lane streams: FETCH_SIZE reads half the bytes there).  Each kernel streams N bytes once in and once out.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T> __global__ void k_copy(const T* __restrict__ a, T* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main()
{
    const size_t bytes = 1ull << 30;   // 1 GiB: far beyond the 256 MiB Infinity Cache
    void *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    (void)hipMemset(a, 1, bytes);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_copy<unsigned char>, dim3(8192), dim3(256), 0, 0, (const unsigned char*)a, (unsigned char*)b, bytes);
        hipLaunchKernelGGL(k_copy<unsigned short>, dim3(8192), dim3(256), 0, 0, (const unsigned short*)a, (unsigned short*)b, bytes / 2);
        hipLaunchKernelGGL(k_copy<unsigned int>, dim3(8192), dim3(256), 0, 0, (const unsigned int*)a, (unsigned int*)b, bytes / 4);
        hipLaunchKernelGGL(k_copy<uint2>, dim3(8192), dim3(256), 0, 0, (const uint2*)a, (uint2*)b, bytes / 8);
        hipLaunchKernelGGL(k_copy<uint4>, dim3(8192), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, bytes / 16);
    }
    (void)hipDeviceSynchronize();
    printf("copied %zu bytes per kernel\n", bytes);
    return 0;
}
