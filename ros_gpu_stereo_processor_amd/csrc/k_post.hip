// k_post.hip -- everything downstream of the matcher on gfx950:
//   validateDisparity (L/R consistency) + ROI blanking, disparity -> float, speckle filter (parallel union-find CCL),
//   reprojectImageTo3D, PointCloud2 packing (also fused disparity -> cloud), drawColorDisp, and the
//   cv::StereoBM::compute wrapper (prefilter + match + finalize).
//
// Reference behaviour being replaced (file:line relative to the reference tree):
//   cv::StereoBM::compute tail            src/GPUStereoProcessor.cpp:319      (validateDisparity, ROI blanking)
//   convertTo(CV_32F, 1/16, -(cx_l-cx_r)) src/GPUStereoProcessor.cpp:320
//   filterSpeckles                         src/GPUStereoProcessor.cpp:356-385  (synchronous CPU flood fill there)
//   projectDisparityTo3DPoints             src/GPUStereoProcessor.cpp:332-346
//   GPUSenderPc2::fillInData               src/GpuSenderPc2.cpp:15-89
//   computeDisparityImage                  src/GPUStereoProcessor.cpp:323-330
// These are HBM-bound passes (the cloud writes 32 B/px); integer results are bit-exact with the oracle.
#include "gip_internal.h"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;
constexpr int TPB = 256;

// ---------------------------------------------------------------------------------------- finalize
// One workgroup per image row.  disp2/cost2 of cv::validateDisparity become one 64-bit key per target column
// (cost << 32 | x << 16 | d) reduced with an LDS atomic min: lowest cost wins, ties go to the smallest x --
// exactly the strict '>' replacement rule of the sequential scan.
__global__ void k_bm_finalize(int16_t* disp, size_t dpitch, const int32_t* cost, size_t cpitch, int w, int h, int mindisp, int ndisp,
                              int disp12, int roi_x, int roi_y, int roi_w, int roi_h, int wr0, int wr1, size_t dbs, size_t cbs)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    u64* keys = (u64*)smem_raw;
    const int y = blockIdx.x;
    int16_t* dptr = (int16_t*)((uint8_t*)disp + blockIdx.z * dbs) + (size_t)y * dpitch;
    const int INVALID = (mindisp - 1) * 16;
    if (y < roi_y || y >= roi_y + roi_h) {
        for (int x = threadIdx.x; x < w; x += blockDim.x) dptr[x] = (int16_t)INVALID;
        return;
    }
    if (disp12 >= 0) {
        const int32_t* cptr = (const int32_t*)((const uint8_t*)cost + blockIdx.z * cbs) + (size_t)y * cpitch;
        const int minD = mindisp, maxD = mindisp + ndisp;
        const int minX1 = maxD > 0 ? maxD : 0, maxX1 = w + (minD < 0 ? minD : 0);
        const int maxdiff = disp12 * 16;
        for (int x = threadIdx.x; x < w; x += blockDim.x) keys[x] = ~0ull;
        __syncthreads();
        for (int x = minX1 + threadIdx.x; x < maxX1; x += blockDim.x) {
            int d = dptr[x];
            if (d == INVALID) continue;
            int x2 = x - ((d + 8) >> 4);
            if (x2 < 0 || x2 >= w) continue;
            u64 key = ((u64)(u32)cptr[x] << 32) | ((u64)(u32)x << 16) | (u64)(unsigned short)d;
            atomicMin(&keys[x2], key);
        }
        __syncthreads();
        for (int x = minX1 + threadIdx.x; x < maxX1; x += blockDim.x) {
            int d = dptr[x];
            if (d == INVALID) continue;
            int d0 = d >> 4, d1 = (d + 15) >> 4;
            int x0 = x - d0, x1 = x - d1;
            bool bad0 = false, bad1 = false;
            if (x0 >= 0 && x0 < w) {
                u64 k = keys[x0];
                if (k != ~0ull) { int d2 = (int)(short)(unsigned short)(k & 0xFFFFu); int df = d2 - d; bad0 = (df < 0 ? -df : df) > maxdiff; }
            }
            if (x1 >= 0 && x1 < w) {
                u64 k = keys[x1];
                if (k != ~0ull) { int d2 = (int)(short)(unsigned short)(k & 0xFFFFu); int df = d2 - d; bad1 = (df < 0 ? -df : df) > maxdiff; }
            }
            if (bad0 && bad1) dptr[x] = (int16_t)INVALID;
        }
    }
    // blank everything outside the valid ROI columns (after the L/R check, as OpenCV does) and the columns the matcher
    // never writes ([wr0, wr1) = [lofs, lofs + width1) is its output range)
    for (int x = threadIdx.x; x < w; x += blockDim.x)
        if (x < roi_x || x >= roi_x + roi_w || x < wr0 || x >= wr1) dptr[x] = (int16_t)INVALID;
}

__global__ void k_disp_to_f32(const int16_t* __restrict__ disp, size_t dpitch, float* __restrict__ dst, size_t fpitch, int w, int h, float alpha,
                              float beta, size_t dbs, size_t fbs)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    float v = (float)((const int16_t*)((const uint8_t*)disp + blockIdx.z * dbs))[(size_t)y * dpitch + x];
    v = v * alpha;
    v = v + beta;
    ((float*)((uint8_t*)dst + blockIdx.z * fbs))[(size_t)y * fpitch + x] = v;
}

// ---------------------------------------------------------------------------------------- speckle filter
// cv::filterSpeckles = remove 4-connected components (neighbours joined when both != newVal and |diff| <= maxDiff) of
// size <= maxSpeckleSize.  Components of a fixed symmetric graph => order independent => a parallel union-find gives
// the same result bit for bit.
//
// Run-based CCL (the union-find nodes are horizontal RUNS, not pixels, so the number of global atomics scales with the
// number of run overlaps instead of the number of pixels):
//   k_spk_runs   one workgroup per image row: every valid pixel gets the COLUMN of the first pixel of its run (u16, the only
//                dense scratch array; segmented max-scan in LDS); run starts initialise their union-find node (parent =
//                self, size 0); the last pixel of a run stores the run length at the start
//   k_spk_merge  one union per maximal overlap segment between runs of adjacent rows (atomicMin union-find on run starts)
//   k_spk_sizes  every run start: root = find(); atomicAdd(size[root], run length); path compression
//   k_spk_apply  pixels whose component has <= maxSpeckleSize pixels become newVal
__device__ __forceinline__ int uf_find(int* lab, int i)
{
    // agent-scope relaxed loads: other workgroups (other XCDs) relink concurrently; a stale parent is still an
    // ancestor, so the loop terminates, and the union below only trusts what the atomic returns.
    int p = __hip_atomic_load(&lab[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != i) {
        i = p;
        p = __hip_atomic_load(&lab[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return i;
}

__device__ __forceinline__ void uf_union(int* lab, int a, int b)
{
    bool done;
    do {
        a = uf_find(lab, a);
        b = uf_find(lab, b);
        if (a < b) {
            int old = atomicMin(&lab[b], a);
            done = (old == b);
            b = old;
        } else if (b < a) {
            int old = atomicMin(&lab[a], b);
            done = (old == a);
            a = old;
        } else {
            done = true;
        }
    } while (!done);
}

__device__ __forceinline__ bool spk_conn(int a, int b, int new_val, int max_diff)
{
    int df = a - b;
    return a != new_val && b != new_val && (df < 0 ? -df : df) <= max_diff;
}

constexpr int SPK_TPB = 256;

// Scratch per pixel: start16 (u16: column of the pixel's run start, 0xFFFF = invalid pixel) is the only DENSE array;
// parent / rlen / csize (int32) are touched only at run-start pixels.
__global__ void __launch_bounds__(SPK_TPB) k_spk_runs(const int16_t* __restrict__ disp, size_t dpitch, unsigned short* __restrict__ start16,
                                                       int* __restrict__ parent, int* __restrict__ rlen, int* __restrict__ csize, int w, int h, int new_val,
                                                       int max_diff, size_t dbs)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    short* row = (short*)smem_raw;                 // w values (+1 pad)
    int* part = (int*)(smem_raw + (((size_t)w * 2 + 2 + 15) & ~(size_t)15));  // SPK_TPB partial maxima
    const int y = blockIdx.x, t = threadIdx.x;
    const int16_t* d = (const int16_t*)((const uint8_t*)disp + blockIdx.z * dbs) + (size_t)y * dpitch;
    const size_t base = blockIdx.z * (size_t)w * h + (size_t)y * w;
    for (int x = t; x < w; x += SPK_TPB) row[x] = d[x];
    __syncthreads();
    const int chunk = (w + SPK_TPB - 1) / SPK_TPB;
    const int x0 = t * chunk, x1 = min(x0 + chunk, w);
    // pass 1: last run start inside my chunk
    int last = -1;
    for (int x = x0; x < x1; x++) {
        int v = row[x];
        bool start = v != new_val && (x == 0 || !spk_conn(row[x - 1], v, new_val, max_diff));
        if (start) last = x;
    }
    part[t] = last;
    __syncthreads();
    // inclusive max-scan over the partials (Hillis-Steele, 8 steps)
    for (int off = 1; off < SPK_TPB; off <<= 1) {
        int v = part[t];
        int o = t >= off ? part[t - off] : -1;
        __syncthreads();
        part[t] = v > o ? v : o;
        __syncthreads();
    }
    int running = t > 0 ? part[t - 1] : -1;
    // pass 2: run start of every pixel; run starts initialise their union-find node, the run's last pixel stores its length
    for (int x = x0; x < x1; x++) {
        int v = row[x];
        bool valid = v != new_val;
        bool start = valid && (x == 0 || !spk_conn(row[x - 1], v, new_val, max_diff));
        if (start) {
            running = x;
            parent[base + x] = y * w + x;
            csize[base + x] = 0;
        }
        start16[base + x] = valid ? (unsigned short)running : (unsigned short)0xFFFFu;
        bool last_of_run = valid && (x == w - 1 || !spk_conn(v, row[x + 1], new_val, max_diff));
        if (last_of_run) rlen[base + running] = x - running + 1;
    }
}

__global__ void k_spk_merge(const int16_t* __restrict__ disp, size_t dpitch, const unsigned short* __restrict__ start16, int* __restrict__ parent,
                            int w, int h, int new_val, int max_diff, size_t dbs)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y + 1 >= h) return;
    const int16_t* d0 = (const int16_t*)((const uint8_t*)disp + blockIdx.z * dbs) + (size_t)y * dpitch;
    const int16_t* d1 = d0 + dpitch;
    int a = d0[x], b = d1[x];
    if (!spk_conn(a, b, new_val, max_diff)) return;
    if (x > 0) {
        // the pair (x-1) already joins the same two runs: this vertical edge is redundant
        int al = d0[x - 1], bl = d1[x - 1];
        if (spk_conn(al, a, new_val, max_diff) && spk_conn(bl, b, new_val, max_diff) && spk_conn(al, bl, new_val, max_diff)) return;
    }
    const size_t base = blockIdx.z * (size_t)w * h;
    const int i = y * w + x;
    uf_union(parent + base, y * w + (int)start16[base + i], (y + 1) * w + (int)start16[base + i + w]);
}

__global__ void k_spk_sizes(const unsigned short* __restrict__ start16, int* __restrict__ parent, const int* __restrict__ rlen, int* __restrict__ csize,
                            int w, int h)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const size_t base = blockIdx.z * (size_t)w * h;
    const int i = y * w + x;
    if ((int)start16[base + i] != x) return;   // not a run start (0xFFFF never equals a column: w <= 32768)
    int root = parent[base + i];
    int q = parent[base + root];
    while (q != root) { root = q; q = parent[base + root]; }   // the merge kernel is complete: plain loads are coherent
    parent[base + i] = root;                                    // compression: starts now point straight at their root
    atomicAdd(&csize[base + root], rlen[base + i]);
}

__global__ void k_spk_apply(int16_t* __restrict__ disp, size_t dpitch, const unsigned short* __restrict__ start16, const int* __restrict__ parent,
                            const int* __restrict__ csize, int w, int h, int new_val, int max_size, size_t dbs)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const size_t base = blockIdx.z * (size_t)w * h;
    const unsigned s = start16[base + y * w + x];
    if (s == 0xFFFFu) return;
    const int root = parent[base + y * w + (int)s];   // a start's entry holds its root (k_spk_sizes compressed it)
    if (csize[base + root] <= max_size) ((int16_t*)((uint8_t*)disp + blockIdx.z * dbs))[(size_t)y * dpitch + x] = (int16_t)new_val;
}

// ---------------------------------------------------------------------------------------- reprojection / cloud
struct QMat { double q[16]; };

// cv::reprojectImageTo3D on a float disparity (double arithmetic, float store).  OpenCV accumulates qx += Q00 per
// column; evaluating Q00*x directly differs by at most the last double bit, i.e. <= 1 float ulp after the cast.
__global__ void k_reproject(const float* __restrict__ disp, size_t dpitch, float* __restrict__ xyz, size_t xpitch, int w, int h, QMat Q,
                            float missing, int handle_missing, size_t dbs, size_t xbs)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const double* q = Q.q;
    float df = ((const float*)((const uint8_t*)disp + blockIdx.z * dbs))[(size_t)y * dpitch + x];
    double d = df;
    double qx = q[1] * y + q[3] + q[0] * x, qy = q[5] * y + q[7] + q[4] * x;
    double qz = q[9] * y + q[11] + q[8] * x, qw = q[13] * y + q[15] + q[12] * x;
    double iW = 1. / (qw + q[14] * d);
    double X = (qx + q[2] * d) * iW, Y = (qy + q[6] * d) * iW, Z = (qz + q[10] * d) * iW;
    if (handle_missing && fabs(d - (double)missing) <= 1.1920928955078125e-07) Z = 10000.;
    float* o = (float*)((uint8_t*)xyz + blockIdx.z * xbs) + (size_t)y * xpitch + 3 * (size_t)x;
    o[0] = (float)X; o[1] = (float)Y; o[2] = (float)Z;
}

__device__ __forceinline__ void store_pc2_record(uint8_t* rec, float X, float Y, float Z, u32 bgr)
{
    const bool valid = (Z != 10000.f) && !isinf(Z);
    const float nanv = __builtin_nanf("");
    float4 a = make_float4(valid ? X : nanv, valid ? Y : nanv, valid ? Z : nanv, 0.f);
    uint4 b = make_uint4(bgr, 0u, 0u, 0u);
    // write-once stream (0.95 GB per 64 frames): non-temporal stores keep it from displacing the tables / images in L2
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store((f4){a.x, a.y, a.z, a.w}, (f4*)rec);          // x y z + 4 pad bytes (0)
    __builtin_nontemporal_store((u4){b.x, b.y, b.z, b.w}, (u4*)rec + 1);      // rgb (b,g,r,0) + 12 pad bytes (0)
}

__global__ void k_pack_pc2(const float* __restrict__ xyz, size_t xpitch, const uint8_t* __restrict__ bgr, size_t bpitch, uint8_t* __restrict__ out,
                           int w, int h, size_t xbs, size_t bbs, size_t obs)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const float* p = (const float*)((const uint8_t*)xyz + blockIdx.z * xbs) + (size_t)y * xpitch + 3 * (size_t)x;
    const uint8_t* c = bgr + blockIdx.z * bbs + (size_t)y * bpitch + 3 * (size_t)x;
    u32 col = (u32)c[0] | ((u32)c[1] << 8) | ((u32)c[2] << 16);
    store_pc2_record(out + blockIdx.z * obs + ((size_t)y * w + x) * 32, p[0], p[1], p[2], col);
}

// fused: int16 x16 disparity -> float (d/16 - (cx_l-cx_r)) [optionally stored] -> reproject -> 32-byte record.
// Missing = FILTERED, which is always the image minimum of a StereoBM map (rows 0..r-1 are FILTERED and every valid value
// is >= mindisp*16).  CN = channels of the colour image (1: mono replicated = GRAY2BGR, 3: BGR).
template <int CN>
__global__ void k_disp_outputs(const int16_t* __restrict__ disp, size_t dpitch, const uint8_t* __restrict__ col, size_t cpitch,
                               float* __restrict__ f32out, uint8_t* __restrict__ pc, int w, int h, QMat Q, float alpha, float beta, int filtered,
                               size_t dbs, size_t cbs, size_t fbs, size_t obs)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const double* q = Q.q;
    int d16 = ((const int16_t*)((const uint8_t*)disp + blockIdx.z * dbs))[(size_t)y * dpitch + x];
    float df = (float)d16 * alpha;
    df = df + beta;
    if (f32out) ((float*)((uint8_t*)f32out + blockIdx.z * fbs))[(size_t)y * w + x] = df;
    if (!pc) return;
    double d = df;
    double qx = q[1] * y + q[3] + q[0] * x, qy = q[5] * y + q[7] + q[4] * x;
    double qz = q[9] * y + q[11] + q[8] * x, qw = q[13] * y + q[15] + q[12] * x;
    double iW = 1. / (qw + q[14] * d);
    double X = (qx + q[2] * d) * iW, Y = (qy + q[6] * d) * iW, Z = (qz + q[10] * d) * iW;
    if (d16 == filtered) Z = 10000.;
    u32 c = 0;
    if (col) {
        const uint8_t* cp = col + blockIdx.z * cbs + (size_t)y * cpitch + (size_t)CN * x;
        if (CN == 1) c = (u32)cp[0] * 0x010101u;
        else c = (u32)cp[0] | ((u32)cp[1] << 8) | ((u32)cp[2] << 16);
    }
    store_pc2_record(pc + blockIdx.z * obs + ((size_t)y * w + x) * 32, (float)X, (float)Y, (float)Z, c);
}

// cv::cuda::drawColorDisp: HSV ramp (S = V = 1) of the integer disparity; invalid (< 0) -> 0
__global__ void k_draw_color_disp(const int16_t* __restrict__ disp, size_t dpitch, uint8_t* __restrict__ bgra, size_t bpitch, int w, int h, int ndisp,
                                  size_t dbs, size_t bbs)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    int d16 = ((const int16_t*)((const uint8_t*)disp + blockIdx.z * dbs))[(size_t)y * dpitch + x];
    int d = d16 < 0 ? 0 : d16 >> 4;
    unsigned H = (unsigned)(((ndisp - d) * 240) / ndisp);
    unsigned hi = (H / 60) % 6;
    float f = H / 60.f - (float)(H / 60);
    float p = 0.f, qv = 1.f - f, t = 1.f - (1.f - f), V = 1.f;
    float b, g, r;
    switch (hi) {
    case 0: b = p; g = t; r = V; break;
    case 1: b = p; g = V; r = qv; break;
    case 2: b = t; g = V; r = p; break;
    case 3: b = V; g = qv; r = p; break;
    case 4: b = V; g = p; r = t; break;
    default: b = qv; g = p; r = V; break;
    }
    u32 bb = (u32)(fmaxf(0.f, fminf(b, 1.f)) * 255.f), gg = (u32)(fmaxf(0.f, fminf(g, 1.f)) * 255.f), rr = (u32)(fmaxf(0.f, fminf(r, 1.f)) * 255.f);
    ((u32*)(bgra + blockIdx.z * bbs + (size_t)y * bpitch))[x] = bb | (gg << 8) | (rr << 16) | (255u << 24);
}

inline dim3 grid2d(int w, int h, int batch) { return dim3(gip_div_up(w, TPB), h, batch); }

}  // namespace

#define GIPK_ARGCHECK(cond) do { if (!(cond)) return GIP_ERR_INVALID_ARG; } while (0)

extern "C" {

int gipk_bm_finalize(int16_t* disp, size_t disp_pitch, const int32_t* cost, size_t cost_pitch, int w, int h, const gip_bm_params* p, int batch,
                     size_t disp_bstride, size_t cost_bstride, gip_stream stream)
{
    GIPK_ARGCHECK(disp && p && w > 0 && h > 0 && batch > 0 && disp_pitch % 2 == 0);
    GIPK_ARGCHECK(p->disp12_max_diff < 0 || (cost && cost_pitch % 4 == 0));
    if (w > 16384) return GIP_ERR_UNSUPPORTED;
    const GipBmGeom g = gip_bm_geom(w, h, p);
    size_t sh = p->disp12_max_diff >= 0 ? (size_t)w * 8 : 16;
    if (sh > 64 * 1024) GIP_HIP_CHECK(hipFuncSetAttribute((const void*)k_bm_finalize, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    int rx = g.roi_x, ry = g.roi_y, rw = g.roi_w, rh = g.roi_h;
    if (g.degenerate) { rx = ry = rw = rh = 0; }
    hipLaunchKernelGGL(k_bm_finalize, dim3(h, 1, batch), dim3(TPB), sh, (hipStream_t)stream, disp, disp_pitch / 2, cost, cost ? cost_pitch / 4 : 0, w, h,
                       p->min_disparity, p->num_disparities, g.degenerate ? -1 : p->disp12_max_diff, rx, ry, rw, rh, g.lofs, g.lofs + g.width1, disp_bstride,
                       cost_bstride);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_disparity_to_f32(const int16_t* disp, size_t disp_pitch, float* dst, size_t dst_pitch, int w, int h, double cx_l, double cx_r, int batch,
                          size_t disp_bstride, size_t dst_bstride, gip_stream stream)
{
    GIPK_ARGCHECK(disp && dst && w > 0 && h > 0 && batch > 0 && disp_pitch % 2 == 0 && dst_pitch % 4 == 0);
    hipLaunchKernelGGL(k_disp_to_f32, grid2d(w, h, batch), dim3(TPB), 0, (hipStream_t)stream, disp, disp_pitch / 2, dst, dst_pitch / 4, w, h,
                       (float)(1.0 / 16), (float)(-(cx_l - cx_r)), disp_bstride, dst_bstride);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

size_t gipk_speckle_scratch_bytes(int w, int h, int batch) { return (size_t)w * h * batch * 14 + 64; }

int gipk_filter_speckles(int16_t* disp, size_t disp_pitch, int w, int h, int new_val, int max_speckle_size, int max_diff, void* scratch,
                         size_t scratch_bytes, int batch, size_t disp_bstride, gip_stream stream)
{
    GIPK_ARGCHECK(disp && scratch && w > 0 && h > 0 && batch > 0 && disp_pitch % 2 == 0);
    GIPK_ARGCHECK(scratch_bytes >= gipk_speckle_scratch_bytes(w, h, batch) && (size_t)w * h < (1u << 31));
    if (w > 32768) return GIP_ERR_UNSUPPORTED;
    const size_t n = (size_t)w * h * batch;
    int* parent = (int*)gip_align_up((size_t)(uintptr_t)scratch, 16);
    int* rlen = parent + n;
    int* csize = rlen + n;
    unsigned short* start16 = (unsigned short*)(csize + n);
    hipStream_t st = (hipStream_t)stream;
    dim3 g = grid2d(w, h, batch);
    size_t sh = (((size_t)w * 2 + 2 + 15) & ~(size_t)15) + SPK_TPB * sizeof(int);
    if (sh > 64 * 1024) GIP_HIP_CHECK(hipFuncSetAttribute((const void*)k_spk_runs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    hipLaunchKernelGGL(k_spk_runs, dim3(h, 1, batch), dim3(SPK_TPB), sh, st, disp, disp_pitch / 2, start16, parent, rlen, csize, w, h, new_val, max_diff, disp_bstride);
    hipLaunchKernelGGL(k_spk_merge, g, dim3(TPB), 0, st, disp, disp_pitch / 2, start16, parent, w, h, new_val, max_diff, disp_bstride);
    hipLaunchKernelGGL(k_spk_sizes, g, dim3(TPB), 0, st, start16, parent, rlen, csize, w, h);
    hipLaunchKernelGGL(k_spk_apply, g, dim3(TPB), 0, st, disp, disp_pitch / 2, start16, parent, csize, w, h, new_val, max_speckle_size, disp_bstride);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_reproject_to_3d(const float* disp, size_t disp_pitch, float* xyz, size_t xyz_pitch, int w, int h, const double Q[16], float missing_value,
                         int handle_missing, int batch, size_t disp_bstride, size_t xyz_bstride, gip_stream stream)
{
    GIPK_ARGCHECK(disp && xyz && Q && w > 0 && h > 0 && batch > 0 && disp_pitch % 4 == 0 && xyz_pitch % 4 == 0);
    QMat q;
    for (int i = 0; i < 16; i++) q.q[i] = Q[i];
    hipLaunchKernelGGL(k_reproject, grid2d(w, h, batch), dim3(TPB), 0, (hipStream_t)stream, disp, disp_pitch / 4, xyz, xyz_pitch / 4, w, h, q, missing_value,
                       handle_missing, disp_bstride, xyz_bstride);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_pack_pointcloud2(const float* xyz, size_t xyz_pitch, const uint8_t* bgr, size_t bgr_pitch, uint8_t* out, int w, int h, int batch,
                          size_t xyz_bstride, size_t bgr_bstride, size_t out_bstride, gip_stream stream)
{
    GIPK_ARGCHECK(xyz && bgr && out && w > 0 && h > 0 && batch > 0 && xyz_pitch % 4 == 0 && ((uintptr_t)out & 15) == 0 && out_bstride % 16 == 0);
    hipLaunchKernelGGL(k_pack_pc2, grid2d(w, h, batch), dim3(TPB), 0, (hipStream_t)stream, xyz, xyz_pitch / 4, bgr, bgr_pitch, out, w, h, xyz_bstride,
                       bgr_bstride, out_bstride);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_disparity_outputs(const int16_t* disp, size_t disp_pitch, const uint8_t* color, size_t color_pitch, int color_cn, float* f32_out,
                           uint8_t* pc_out, int w, int h, const double Q[16], double cx_l, double cx_r, int min_disparity, int batch,
                           size_t disp_bstride, size_t color_bstride, gip_stream stream)
{
    GIPK_ARGCHECK(disp && Q && (f32_out || pc_out) && w > 0 && h > 0 && batch > 0 && disp_pitch % 2 == 0);
    GIPK_ARGCHECK(!pc_out || ((uintptr_t)pc_out & 15) == 0);
    GIPK_ARGCHECK(!color || color_cn == 1 || color_cn == 3);
    QMat q;
    for (int i = 0; i < 16; i++) q.q[i] = Q[i];
    const float alpha = (float)(1.0 / 16), beta = (float)(-(cx_l - cx_r));
    const size_t npx = (size_t)w * h;
    if (color && color_cn == 1)
        hipLaunchKernelGGL(k_disp_outputs<1>, grid2d(w, h, batch), dim3(TPB), 0, (hipStream_t)stream, disp, disp_pitch / 2, color, color_pitch, f32_out, pc_out, w, h, q,
                           alpha, beta, (min_disparity - 1) * 16, disp_bstride, color_bstride, npx * 4, npx * 32);
    else
        hipLaunchKernelGGL(k_disp_outputs<3>, grid2d(w, h, batch), dim3(TPB), 0, (hipStream_t)stream, disp, disp_pitch / 2, color, color_pitch, f32_out, pc_out, w, h, q,
                           alpha, beta, (min_disparity - 1) * 16, disp_bstride, color_bstride, npx * 4, npx * 32);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_disparity_to_pointcloud2(const int16_t* disp, size_t disp_pitch, const uint8_t* bgr, size_t bgr_pitch, uint8_t* out, int w, int h,
                                  const double Q[16], double cx_l, double cx_r, int min_disparity, int batch, size_t disp_bstride, size_t bgr_bstride,
                                  size_t out_bstride, gip_stream stream)
{
    GIPK_ARGCHECK(out && out_bstride == (size_t)w * h * 32);
    return gipk_disparity_outputs(disp, disp_pitch, bgr, bgr_pitch, 3, nullptr, out, w, h, Q, cx_l, cx_r, min_disparity, batch, disp_bstride, bgr_bstride, stream);
}

int gipk_draw_color_disp(const int16_t* disp, size_t disp_pitch, uint8_t* bgra, size_t bgra_pitch, int w, int h, int ndisp, int batch,
                         size_t disp_bstride, size_t bgra_bstride, gip_stream stream)
{
    GIPK_ARGCHECK(disp && bgra && w > 0 && h > 0 && batch > 0 && ndisp > 0 && disp_pitch % 2 == 0 && bgra_pitch % 4 == 0 && ((uintptr_t)bgra & 3) == 0);
    hipLaunchKernelGGL(k_draw_color_disp, grid2d(w, h, batch), dim3(TPB), 0, (hipStream_t)stream, disp, disp_pitch / 2, bgra, bgra_pitch, w, h, ndisp,
                       disp_bstride, bgra_bstride);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

// scratch layout of gipk_stereobm: [prefiltered L][prefiltered R][cost plane][speckle labels+sizes]
static size_t bm_plane_pitch(int w) { return gip_align_up((size_t)w, 64); }

size_t gipk_stereobm_scratch_bytes(int w, int h, int batch, const gip_bm_params* p)
{
    size_t pp = bm_plane_pitch(w);
    size_t bytes = 2 * pp * h * batch + 256;
    bytes += (size_t)w * h * batch * 4 + 256;
    if (p && p->speckle_range >= 0 && p->speckle_window_size > 0) bytes += gipk_speckle_scratch_bytes(w, h, batch) + 256;
    return bytes;
}

int gipk_stereobm(const uint8_t* left, const uint8_t* right, size_t pitch, int w, int h, const gip_bm_params* p, int16_t* disp, size_t disp_pitch,
                  void* scratch, size_t scratch_bytes, int batch, size_t img_bstride, size_t disp_bstride, gip_stream stream)
{
    GIPK_ARGCHECK(left && right && disp && scratch && p && w > 0 && h > 0 && batch > 0);
    int rc = gip_validate_bm_params(p, w, h);
    if (rc != GIP_OK) return rc;
    if (scratch_bytes < gipk_stereobm_scratch_bytes(w, h, batch, p)) return GIP_ERR_INVALID_ARG;
    const size_t pp = bm_plane_pitch(w), plane = pp * h;
    uint8_t* base = (uint8_t*)gip_align_up((size_t)(uintptr_t)scratch, 256);
    uint8_t* lf = base;
    uint8_t* rf = lf + plane * batch;
    int32_t* cost = (int32_t*)gip_align_up((size_t)(uintptr_t)(rf + plane * batch), 256);
    void* spk = (void*)gip_align_up((size_t)(uintptr_t)((uint8_t*)cost + (size_t)w * h * batch * 4), 256);
    rc = gipk_prefilter(left, pitch, lf, pp, w, h, p->prefilter_type, p->prefilter_size, p->prefilter_cap, batch, img_bstride, plane, stream);
    if (rc != GIP_OK) return rc;
    rc = gipk_prefilter(right, pitch, rf, pp, w, h, p->prefilter_type, p->prefilter_size, p->prefilter_cap, batch, img_bstride, plane, stream);
    if (rc != GIP_OK) return rc;
    const bool lr = p->disp12_max_diff >= 0;
    rc = gip_bm_match_impl(lf, rf, pp, w, h, p, disp, disp_pitch, lr ? cost : nullptr, (size_t)w * 4, batch, plane, disp_bstride, (size_t)w * h * 4, stream, true);
    if (rc != GIP_OK) return rc;
    rc = gipk_bm_finalize(disp, disp_pitch, lr ? cost : nullptr, (size_t)w * 4, w, h, p, batch, disp_bstride, (size_t)w * h * 4, stream);
    if (rc != GIP_OK) return rc;
    if (p->speckle_range >= 0 && p->speckle_window_size > 0)
        rc = gipk_filter_speckles(disp, disp_pitch, w, h, (p->min_disparity - 1) * 16, p->speckle_window_size, p->speckle_range, spk,
                                  gipk_speckle_scratch_bytes(w, h, batch), batch, disp_bstride, stream);
    return rc;
}

}  // extern "C"
