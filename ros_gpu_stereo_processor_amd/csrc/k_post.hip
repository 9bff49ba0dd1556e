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
#include "bm_common.h"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;
constexpr int TPB = 256;

// ---------------------------------------------------------------------------------------- finalize
// One workgroup per image row.  disp2/cost2 of cv::validateDisparity become one 64-bit key per target column
// (cost << 32 | x << 16 | d) reduced with an LDS atomic min: lowest cost wins, ties go to the smallest x --
// exactly the strict '>' replacement rule of the sequential scan.
template <class CT>
__global__ void k_bm_finalize(int16_t* disp, size_t dpitch, const CT* cost, size_t cpitch, int w, int h, int mindisp, int ndisp,
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
        const CT* cptr = (const CT*)((const uint8_t*)cost + blockIdx.z * cbs) + (size_t)y * cpitch;
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

// Same, for rows of at most 256 * NI pixels: every thread keeps its NI pixels in registers across the three phases, so the
// disparity row is read once and only the pixels that change are written back.
// K32 (uint16 costs of the column-sum matcher, ndisp <= 256; used for rows wider than 1536 pixels): the key is ONE dword, cost << 12 | (d - minD * 16).
// For a fixed target column the candidates' source columns x = x2 + ((d + 8) >> 4) are ordered like their disparities, so "smallest d" breaks cost
// ties exactly as "smallest x" does, and d is all the second pass reads back.  Half the LDS: a 4096-pixel row needs 16 KB instead of 32 KB, which is
// the difference between fitting next to a CU's matcher workgroups and waiting for one of them to leave (the L/R check of a 4096x2160 frame took
// 296 us next to the following frame's matcher, 60 us alone).  For narrow rows the 64-bit form stays: there the smaller footprint lets MORE of these
// workgroups in next to the matcher and the pipelined step gets slower.
template <int NI, class CT, bool K32>
__global__ void __launch_bounds__(TPB) k_bm_finalize_r(int16_t* disp, size_t dpitch, const CT* cost, size_t cpitch, int w, int h, int mindisp,
                                                        int ndisp, int disp12, int roi_x, int roi_y, int roi_w, int roi_h, int wr0, int wr1, size_t dbs,
                                                        size_t cbs)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    u64* keys = (u64*)smem_raw;
    u32* keys32 = (u32*)smem_raw;
    const int y = blockIdx.x, t = threadIdx.x;
    int16_t* dptr = (int16_t*)((uint8_t*)disp + blockIdx.z * dbs) + (size_t)y * dpitch;
    const int INVALID = (mindisp - 1) * 16;
    if (y < roi_y || y >= roi_y + roi_h) {
        for (int x = t; x < w; x += TPB) dptr[x] = (int16_t)INVALID;
        return;
    }
    int d[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int x = t + TPB * i;
        d[i] = x < w ? (int)dptr[x] : INVALID;
    }
    bool kill[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) kill[i] = false;
    if (disp12 >= 0) {
        const CT* cptr = (const CT*)((const uint8_t*)cost + blockIdx.z * cbs) + (size_t)y * cpitch;
        const int minD = mindisp, maxD = mindisp + ndisp;
        const int minX1 = maxD > 0 ? maxD : 0, maxX1 = w + (minD < 0 ? minD : 0);
        const int maxdiff = disp12 * 16;
        // (loaded whether or not the pixel is valid: a load that depended on d[i] would be a second memory round trip behind the first)
        u32 c[NI];
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int x = t + TPB * i;
            c[i] = (x >= minX1 && x < maxX1 && x < w) ? (u32)cptr[x] : 0u;
        }
        const int d_lo = mindisp * 16;
        if (K32) for (int x = t; x < w; x += TPB) keys32[x] = ~0u;
        else for (int x = t; x < w; x += TPB) keys[x] = ~0ull;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int x = t + TPB * i;
            if (x < minX1 || x >= maxX1 || d[i] == INVALID) continue;
            const int x2 = x - ((d[i] + 8) >> 4);
            if (x2 < 0 || x2 >= w) continue;
            if (K32) atomicMin(&keys32[x2], (c[i] << 12) | ((u32)(d[i] - d_lo) & 0xFFFu));
            else atomicMin(&keys[x2], ((u64)c[i] << 32) | ((u64)(u32)x << 16) | (u64)(unsigned short)d[i]);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const int x = t + TPB * i;
            if (x < minX1 || x >= maxX1 || d[i] == INVALID) continue;
            const int x0 = x - (d[i] >> 4), x1 = x - ((d[i] + 15) >> 4);
            bool bad0 = false, bad1 = false;
            if (K32) {
                if (x0 >= 0 && x0 < w) {
                    const u32 k = keys32[x0];
                    if (k != ~0u) { const int df = (int)(k & 0xFFFu) + d_lo - d[i]; bad0 = (df < 0 ? -df : df) > maxdiff; }
                }
                if (x1 >= 0 && x1 < w) {
                    const u32 k = keys32[x1];
                    if (k != ~0u) { const int df = (int)(k & 0xFFFu) + d_lo - d[i]; bad1 = (df < 0 ? -df : df) > maxdiff; }
                }
            } else {
                if (x0 >= 0 && x0 < w) {
                    const u64 k = keys[x0];
                    if (k != ~0ull) { const int df = (int)(short)(unsigned short)(k & 0xFFFFu) - d[i]; bad0 = (df < 0 ? -df : df) > maxdiff; }
                }
                if (x1 >= 0 && x1 < w) {
                    const u64 k = keys[x1];
                    if (k != ~0ull) { const int df = (int)(short)(unsigned short)(k & 0xFFFFu) - d[i]; bad1 = (df < 0 ? -df : df) > maxdiff; }
                }
            }
            kill[i] = bad0 && bad1;
        }
    }
    // L/R failures, everything outside the valid ROI columns (after the L/R check, as OpenCV does) and the columns the
    // matcher never writes ([wr0, wr1) = [lofs, lofs + width1) is its output range)
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int x = t + TPB * i;
        if (x >= w) continue;
        const bool blank = kill[i] || x < roi_x || x >= roi_x + roi_w || x < wr0 || x >= wr1;
        if (blank && d[i] != INVALID) dptr[x] = (int16_t)INVALID;
    }
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
// Run-based, two-level CCL.  The union-find nodes are horizontal RUNS, not pixels; node id = y * w + (ordinal of the run in
// its row).  Only ONE kernel touches every pixel, and most unions happen in LDS:
//   k_spk_scan   one WAVEFRONT per image row, marching over it 64 columns at a time with ballots (no LDS, no barrier, no
//                atomics): run ordinal of every pixel (ord16, u16, the only dense scratch array), the row's run list
//                (start column | length << 16, indexed by ordinal) and the row's vertical joins -- one per maximal overlap
//                segment between a run of row y and a run of row y+1 -- appended to a compact per-row list
//   k_spk_strip  one workgroup per strip of SPK_STRIP rows: union-find over the strip's runs in LDS, component sizes
//                accumulated at the strip-local roots; writes parent (= local root) and size (at local roots, else 0)
//   k_spk_cross  one workgroup per strip: global atomicMin union-find over the joins that cross the strip's lower seam (and, for a strip
//                too dense for the LDS level, over the joins of its interior rows)
//   k_spk_sizes  strip-local roots that lost their root status add their size to the global root (+ path compression)
//   k_spk_apply  one wavefront per row: runs whose component has <= maxSpeckleSize pixels are overwritten with newVal
// Why: only ~1 % of the pixels start a union or close a run, but each of those is a chain of dependent memory round trips.
// With one thread per pixel about every second wavefront holds one such lane and stays resident for microseconds (0.19 ms
// for merge and for sizes on 64 x 1242x375 maps); and thousands of simultaneous global unions build trees as deep as the
// image is tall.  Lists make the slow path dense; the LDS level makes it short.
#ifndef SPK_STRIP_N
#define SPK_STRIP_N 8
#endif
#ifndef SPK_CAP_N
#define SPK_CAP_N 1264
#endif
// Strip height / LDS capacity are sized so that a strip workgroup needs just under 10 KB of LDS (2 x 1264 x 4 + 72 = 10 184 <= 10 240, the room eight
// matcher workgroups of the metric's shape leave on a CU).  With 32-row strips (57 KB) the kernel took 642 us
// next to the matcher of another batch against 29 us alone: the matcher keeps every CU's 160 KB full with 20 KB workgroups that are
// replaced one at a time, so a 57 KB request starves.  At 10 KB it takes 99 us co-scheduled (the ~3x every small kernel stretches by);
// the global merge of the strip seams (k_spk_cross) grows from 17 to 31 us, throughput is unchanged (profiles/r02_stretch.txt).
#ifndef SPK_SCAN4_MIN_W
#define SPK_SCAN4_MIN_W 256     /* rows at least this wide are scanned four pixels per lane (k_spk_scan4), narrower ones two (k_spk_scan2) */
#endif
constexpr int SPK_STRIP = SPK_STRIP_N;        // rows per strip
constexpr int SPK_CAP = SPK_CAP_N;        // runs per strip the LDS level can hold (else the strip is merged globally)
constexpr int SPK_TPB = 256;

__device__ __forceinline__ int uf_find(int* nd, int i)
{
    // agent-scope relaxed loads: other workgroups (other XCDs) relink concurrently; a stale parent is still an
    // ancestor, so the loop terminates, and the union below only trusts what the atomic returns.
    // Path halving on the way up: parents only ever decrease (links go to the smaller index), so lowering nd[i] to its
    // grandparent with an atomic min keeps "nd[i] is an ancestor of i" whatever else happens concurrently.
    int p = __hip_atomic_load(&nd[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != i) {
        const int gp = __hip_atomic_load(&nd[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gp != p) __hip_atomic_fetch_min(&nd[i], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        i = p;
        p = gp;
    }
    return i;
}

__device__ __forceinline__ void uf_union(int* nd, int a, int b)
{
    bool done;
    do {
        a = uf_find(nd, a);
        b = uf_find(nd, b);
        if (a < b) {
            int old = atomicMin(&nd[b], a);
            done = (old == b);
            b = old;
        } else if (b < a) {
            int old = atomicMin(&nd[a], b);
            done = (old == a);
            a = old;
        } else {
            done = true;
        }
    } while (!done);
}

// True in the first lane of every run of consecutive lanes that hold the same key (lanes without `on` break a run).  The union-find
// kernels below use it to drop redundant unions: neighbouring joins of a row mostly connect the same two large components, and
// thousands of atomics on one address serialise (54 us for the border link, 39 us for the border union before this).  Runs, not
// distinct keys: a loop that serves one distinct key at a time serialises rows with many small components (k_spk_cross: 22 -> 53 us).
__device__ __forceinline__ bool first_of_key_run(bool on, u64 key)
{
    const int lane = (int)(threadIdx.x & 63u);
    const u64 prev = ((u64)(u32)__shfl_up((int)(key >> 32), 1) << 32) | (u64)(u32)__shfl_up((int)key, 1);
    const int prev_on = __shfl_up(on ? 1 : 0, 1);
    return on && !(lane > 0 && prev_on && prev == key);
}

// the same on an LDS array (workgroup scope)
__device__ __forceinline__ int lds_find(int* nd, int i)
{
    int p = __hip_atomic_load(&nd[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (p != i) {
        i = p;
        p = __hip_atomic_load(&nd[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return i;
}

__device__ __forceinline__ void lds_union(int* nd, int a, int b)
{
    bool done;
    do {
        a = lds_find(nd, a);
        b = lds_find(nd, b);
        if (a < b) {
            int old = __hip_atomic_fetch_min(&nd[b], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            done = (old == b);
            b = old;
        } else if (b < a) {
            int old = __hip_atomic_fetch_min(&nd[a], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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

// Scratch: parent / csize (int32, indexed by node id, filled from the front of every row's w slots), runlist / joinlist
// (u32, same indexing), ord16 (u16 per pixel), nrun / njoin (per row), global_strip (per strip: 1 = merged globally).
struct SpkScratch {
    int* parent;
    int* csize;
    u32* runlist;
    u32* joinlist;
    unsigned short* ord16;
    int* nrun;
    int* njoin;
    int* global_strip;
};

__global__ void __launch_bounds__(64) k_spk_scan(const int16_t* __restrict__ disp, size_t dpitch, SpkScratch S, int w, int h, int new_val,
                                                  int max_diff, size_t dbs)
{
    const int y = blockIdx.x, lane = threadIdx.x;
    const int16_t* d0 = (const int16_t*)((const uint8_t*)disp + blockIdx.z * dbs) + (size_t)y * dpitch;
    const int16_t* d1 = d0 + dpitch;
    const bool has_next = y + 1 < h;
    const size_t base = blockIdx.z * (size_t)w * h + (size_t)y * w;
    const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);   // lanes <= me
    const unsigned long long lt = le >> 1;                                           // lanes <  me
    int n_run = 0, n_join = 0, carry_start = 0;
    // the five values of a chunk are loaded one iteration ahead: the loop is a chain of ballots, and without the prefetch
    // every iteration would wait for its own round trip to memory
    int nv = new_val, nvl = new_val, nvr = new_val, nb = new_val, nbl = new_val;
    auto load = [&](int c0) {
        const int x = c0 + lane;
        const bool in = x < w;
        nv = in ? d0[x] : new_val;
        nvl = (in && x > 0) ? d0[x - 1] : new_val;
        nvr = (x + 1 < w) ? d0[x + 1] : new_val;
        nb = (in && has_next) ? d1[x] : new_val;
        nbl = (in && has_next && x > 0) ? d1[x - 1] : new_val;
    };
    load(0);
    for (int c0 = 0; c0 < w; c0 += 64) {
        const int x = c0 + lane;
        const bool in = x < w;
        const int v = nv, vl = nvl, vr = nvr, b = nb, bl = nbl;
        if (c0 + 64 < w) load(c0 + 64);
        const bool valid = v != new_val;
        const bool cl = spk_conn(vl, v, new_val, max_diff);   // joined to the left neighbour
        const unsigned long long smask = __ballot(valid && !cl);
        const unsigned long long m = smask & le;
        const int ord = n_run + __popcll(m) - 1;
        const int rs = m ? (c0 + 63 - __builtin_clzll(m)) : carry_start;
        if (in) S.ord16[base + x] = valid ? (unsigned short)ord : (unsigned short)0xFFFFu;
        if (valid && !spk_conn(v, vr, new_val, max_diff))          // last pixel of its run (vr = new_val past the row end)
            S.runlist[base + ord] = (u32)rs | ((u32)(x - rs + 1) << 16);   // length <= w <= 32768
        // vertical join, unless the pair (x-1) already joins the same two runs
        const bool join = spk_conn(v, b, new_val, max_diff) && !(cl && spk_conn(bl, b, new_val, max_diff) && spk_conn(vl, bl, new_val, max_diff));
        const unsigned long long jmask = __ballot(join);
        if (join) S.joinlist[base + n_join + __popcll(jmask & lt)] = (u32)ord | ((u32)x << 16);
        n_join += __popcll(jmask);
        n_run += __popcll(smask);
        carry_start = __shfl(rs, 63);
    }
    if (lane == 0) {
        S.nrun[blockIdx.z * (size_t)h + y] = n_run;
        S.njoin[blockIdx.z * (size_t)h + y] = n_join;
    }
}

// The same scan with TWO pixels per lane (128 columns per step, dword loads and stores): sub-dword accesses cost the memory
// pipeline as much as dwords, so this halves its work.  Needs an even width and 4-byte aligned rows; k_spk_scan is the fallback.
__global__ void __launch_bounds__(64) k_spk_scan2(const int16_t* __restrict__ disp, size_t dpitch, SpkScratch S, int w, int h, int new_val,
                                                   int max_diff, size_t dbs)
{
    const int y = blockIdx.x, lane = threadIdx.x;
    const int16_t* d0 = (const int16_t*)((const uint8_t*)disp + blockIdx.z * dbs) + (size_t)y * dpitch;
    const int16_t* d1 = d0 + dpitch;
    const bool has_next = y + 1 < h;
    const size_t base = blockIdx.z * (size_t)w * h + (size_t)y * w;
    const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);   // lanes <= me
    const unsigned long long lt = le >> 1;                                           // lanes <  me
    const u32 nvp = (u32)(unsigned short)new_val * 0x10001u;                         // two invalid pixels
    auto load2 = [&](const int16_t* row, int x, bool on) -> u32 { return (on && x < w) ? *(const u32*)(row + x) : nvp; };
    int n_run = 0, n_join = 0, carry_start = 0;
    int pv = new_val, pb = new_val;                 // last pixel of the previous step (rows y and y+1)
    u32 nV = load2(d0, 2 * lane, true), nB = load2(d1, 2 * lane, has_next);
    for (int c0 = 0; c0 < w; c0 += 128) {
        const int x0 = c0 + 2 * lane, x1 = x0 + 1;
        const u32 V = nV, B = nB;
        nV = load2(d0, x0 + 128, c0 + 128 < w);
        nB = load2(d1, x0 + 128, has_next && c0 + 128 < w);
        const int v0 = (short)(V & 0xFFFFu), v1 = (short)(V >> 16), b0 = (short)(B & 0xFFFFu), b1 = (short)(B >> 16);
        int vl = __shfl_up(v1, 1), bl = __shfl_up(b1, 1), vr = __shfl_down(v0, 1);
        const int nfirst = (short)((u32)__shfl((int)nV, 0) & 0xFFFFu);
        if (lane == 0) { vl = pv; bl = pb; }
        if (lane == 63) vr = nfirst;
        pv = __shfl(v1, 63);
        pb = __shfl(b1, 63);
        const bool valid0 = v0 != new_val, valid1 = v1 != new_val;
        const bool cl0 = spk_conn(vl, v0, new_val, max_diff), cl1 = spk_conn(v0, v1, new_val, max_diff);
        const bool st0 = valid0 && !cl0, st1 = valid1 && !cl1;
        const unsigned long long s0 = __ballot(st0), s1 = __ballot(st1);
        const unsigned long long m0 = s0 & le, m1 = s1 & lt;
        const int ord0 = n_run + __popcll(m0) + __popcll(m1) - 1, ord1 = ord0 + (st1 ? 1 : 0);
        const int p0 = m0 ? 2 * (63 - __builtin_clzll(m0)) : -1, p1 = m1 ? 2 * (63 - __builtin_clzll(m1)) + 1 : -1;
        const int pm = p0 > p1 ? p0 : p1;
        const int rs0 = pm >= 0 ? c0 + pm : carry_start;
        const int rs1 = st1 ? x1 : rs0;
        if (x0 < w)
            *(u32*)(S.ord16 + base + x0) = (valid0 ? (u32)ord0 : 0xFFFFu) | ((valid1 ? (u32)ord1 : 0xFFFFu) << 16);
        if (valid0 && !cl1) S.runlist[base + ord0] = (u32)rs0 | ((u32)(x0 - rs0 + 1) << 16);           // pixel 0 closes its run
        if (valid1 && !spk_conn(v1, vr, new_val, max_diff)) S.runlist[base + ord1] = (u32)rs1 | ((u32)(x1 - rs1 + 1) << 16);
        // vertical joins, unless the pair one column to the left already joins the same two runs
        const bool j0 = spk_conn(v0, b0, new_val, max_diff) && !(cl0 && spk_conn(bl, b0, new_val, max_diff) && spk_conn(vl, bl, new_val, max_diff));
        const bool j1 = spk_conn(v1, b1, new_val, max_diff) && !(cl1 && spk_conn(b0, b1, new_val, max_diff) && spk_conn(v0, b0, new_val, max_diff));
        const unsigned long long jm0 = __ballot(j0), jm1 = __ballot(j1);
        const int jp = n_join + __popcll(jm0 & lt) + __popcll(jm1 & lt);
        if (j0) S.joinlist[base + jp] = (u32)ord0 | ((u32)x0 << 16);
        if (j1) S.joinlist[base + jp + (j0 ? 1 : 0)] = (u32)ord1 | ((u32)x1 << 16);
        n_join += __popcll(jm0) + __popcll(jm1);
        n_run += __popcll(s0) + __popcll(s1);
        carry_start = __shfl(rs1, 63);
    }
    if (lane == 0) {
        S.nrun[blockIdx.z * (size_t)h + y] = n_run;
        S.njoin[blockIdx.z * (size_t)h + y] = n_join;
    }
}

// The same scan with FOUR pixels per lane (256 columns per step, 8-byte loads and stores): the loop is a chain of ballots and prefix
// counts whose cost per step hardly depends on how many pixels a lane carries, so four pixels per lane halve the steps of a row again
// (a 4096-column row: 16 steps instead of 32; k_spk_scan2 took 30 us for the 540 rows of a quarter of a 4096x2160 frame, one wave per
// row).  Same conditions as k_spk_scan2 (even width, 4-byte aligned rows); a lane whose quad straddles the row end carries invalid pixels.
__global__ void __launch_bounds__(64) k_spk_scan4(const int16_t* __restrict__ disp, size_t dpitch, SpkScratch S, int w, int h, int new_val,
                                                   int max_diff, size_t dbs)
{
    const int y = blockIdx.x, lane = threadIdx.x;
    const int16_t* d0 = (const int16_t*)((const uint8_t*)disp + blockIdx.z * dbs) + (size_t)y * dpitch;
    const int16_t* d1 = d0 + dpitch;
    const bool has_next = y + 1 < h;
    const size_t base = blockIdx.z * (size_t)w * h + (size_t)y * w;
    const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);   // lanes <= me
    const unsigned long long lt = le >> 1;                                           // lanes <  me
    const u32 nvp = (u32)(unsigned short)new_val * 0x10001u;                         // two invalid pixels
    // four pixels from column x (x even): two dwords, each predicated on its own pair lying inside the row
    auto load4 = [&](const int16_t* row, int x, bool on) -> uint2 {
        uint2 v = make_uint2(nvp, nvp);
        if (on && x + 3 < w) v = *(const uint2*)(row + x);          // (4-byte aligned: rows and x are even counts of int16)
        else if (on && x + 1 < w) v.x = *(const u32*)(row + x);
        return v;
    };
    int n_run = 0, n_join = 0, carry_start = 0;
    int pv = new_val, pb = new_val;                 // last pixel of the previous step (rows y and y+1)
    uint2 nV = load4(d0, 4 * lane, true), nB = load4(d1, 4 * lane, has_next);
    // the first pixel of the NEXT step (lane 63's right neighbour) is loaded a step earlier still, by itself: read out of nV it would make this
    // step wait for the loads it has only just issued, and the prefetch would be none
    int nF = 256 < w ? (int)d0[256] : new_val;
    for (int c0 = 0; c0 < w; c0 += 256) {
        const int x0 = c0 + 4 * lane;
        const uint2 V = nV, B = nB;
        const int nfirst = nF;
        nV = load4(d0, x0 + 256, c0 + 256 < w);
        nB = load4(d1, x0 + 256, has_next && c0 + 256 < w);
        nF = c0 + 512 < w ? (int)d0[c0 + 512] : new_val;
        int v[4] = {(short)(V.x & 0xFFFFu), (short)(V.x >> 16), (short)(V.y & 0xFFFFu), (short)(V.y >> 16)};
        int b[4] = {(short)(B.x & 0xFFFFu), (short)(B.x >> 16), (short)(B.y & 0xFFFFu), (short)(B.y >> 16)};
        int vl = __shfl_up(v[3], 1), bl = __shfl_up(b[3], 1), vr = __shfl_down(v[0], 1);
        if (lane == 0) { vl = pv; bl = pb; }
        if (lane == 63) vr = nfirst;
        pv = __shfl(v[3], 63);
        pb = __shfl(b[3], 63);
        bool valid[4], cl[4], st[4], jn[4];
        unsigned long long sm[4], jm[4];
        int below = 0, pm = -1;                     // starts in lower lanes; position (in this step) of the latest of them
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int left = i == 0 ? vl : v[i - 1], lbelow = i == 0 ? bl : b[i - 1];
            valid[i] = v[i] != new_val;
            cl[i] = spk_conn(left, v[i], new_val, max_diff);
            st[i] = valid[i] && !cl[i];
            // vertical join, unless the pair one column to the left already joins the same two runs
            jn[i] = spk_conn(v[i], b[i], new_val, max_diff) && !(cl[i] && spk_conn(lbelow, b[i], new_val, max_diff) && spk_conn(left, lbelow, new_val, max_diff));
            sm[i] = __ballot(st[i]);
            jm[i] = __ballot(jn[i]);
            const unsigned long long m = sm[i] & lt;
            below += __popcll(m);
            const int p = m ? 4 * (63 - __builtin_clzll(m)) + i : -1;
            pm = p > pm ? p : pm;
        }
        int ord[4], rs[4];
        int acc = n_run + below - 1, cur = pm >= 0 ? c0 + pm : carry_start;
        int jp = n_join + __popcll(jm[0] & lt) + __popcll(jm[1] & lt) + __popcll(jm[2] & lt) + __popcll(jm[3] & lt);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (st[i]) { acc++; cur = x0 + i; }
            ord[i] = acc;
            rs[i] = cur;
        }
        if (x0 < w) {
            const u32 o0 = (valid[0] ? (u32)ord[0] : 0xFFFFu) | ((valid[1] ? (u32)ord[1] : 0xFFFFu) << 16);
            const u32 o1 = (valid[2] ? (u32)ord[2] : 0xFFFFu) | ((valid[3] ? (u32)ord[3] : 0xFFFFu) << 16);
            if (x0 + 3 < w) *(uint2*)(S.ord16 + base + x0) = make_uint2(o0, o1);
            else *(u32*)(S.ord16 + base + x0) = o0;                  // (x0 + 1 < w: the width is even)
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int right = i == 3 ? vr : v[i + 1];
            if (valid[i] && !spk_conn(v[i], right, new_val, max_diff)) S.runlist[base + ord[i]] = (u32)rs[i] | ((u32)(x0 + i - rs[i] + 1) << 16);   // closes its run
            if (jn[i]) S.joinlist[base + jp] = (u32)ord[i] | ((u32)(x0 + i) << 16);
            jp += jn[i] ? 1 : 0;
        }
        n_join += __popcll(jm[0]) + __popcll(jm[1]) + __popcll(jm[2]) + __popcll(jm[3]);
        n_run += __popcll(sm[0]) + __popcll(sm[1]) + __popcll(sm[2]) + __popcll(sm[3]);
        carry_start = __shfl(rs[3], 63);
    }
    if (lane == 0) {
        S.nrun[blockIdx.z * (size_t)h + y] = n_run;
        S.njoin[blockIdx.z * (size_t)h + y] = n_join;
    }
}

// last row of a strip whose exclusive prefix offset is <= i  (off[0..R], off[R] = total)
__device__ __forceinline__ int spk_row_of(const int* off, int R, int i)
{
    int lo = 0, hi = R - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= i) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ void __launch_bounds__(SPK_TPB) k_spk_strip(SpkScratch S, int w, int h, int nstrips)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    int* lab = (int*)smem_raw;              // SPK_CAP
    int* lsize = lab + SPK_CAP;             // SPK_CAP
    __shared__ int off[SPK_STRIP + 1];      // runs:  exclusive prefix over the strip's rows
    __shared__ int joff[SPK_STRIP + 1];     // joins between two rows of the strip: same
    const int t = threadIdx.x;
    const int y0 = blockIdx.x * SPK_STRIP;
    const int R = min(SPK_STRIP, h - y0);
    const size_t ibase = blockIdx.z * (size_t)w * h;
    const int* nrun = S.nrun + blockIdx.z * (size_t)h + y0;
    const int* njoin = S.njoin + blockIdx.z * (size_t)h + y0;
    if (t <= SPK_STRIP) {
        off[t] = t < R ? nrun[t] : 0;
        joff[t] = t + 1 < R ? njoin[t] : 0;
    }
    __syncthreads();
    if (t == 0) {
        int acc = 0, jacc = 0;
        for (int i = 0; i <= SPK_STRIP; i++) {
            const int n = off[i], j = joff[i];
            off[i] = acc; joff[i] = jacc;
            acc += n; jacc += j;
        }
    }
    __syncthreads();
    const int T = off[R], J = joff[R];
    // every phase below runs over the flattened (row, item) space so that its global loads are all in flight at once
    if (T > SPK_CAP) {
        // too many runs for the LDS level: every run is its own strip-local root and all joins go to k_spk_cross
        if (t == 0) S.global_strip[blockIdx.z * (size_t)nstrips + blockIdx.x] = 1;
        for (int i = t; i < T; i += SPK_TPB) {
            const int yl = spk_row_of(off, R, i), k = i - off[yl];
            const size_t node = ibase + (size_t)(y0 + yl) * w + k;
            S.parent[node] = (y0 + yl) * w + k;
            S.csize[node] = (int)(S.runlist[node] >> 16);
        }
        return;
    }
    if (t == 0) S.global_strip[blockIdx.z * (size_t)nstrips + blockIdx.x] = 0;
    for (int i = t; i < T; i += SPK_TPB) { lab[i] = i; lsize[i] = 0; }
    __syncthreads();
    // joins inside the strip
    for (int j = t; j < J; j += SPK_TPB) {
        const int yl = spk_row_of(joff, R, j);
        const size_t rb = ibase + (size_t)(y0 + yl) * w;
        const u32 item = S.joinlist[rb + (j - joff[yl])];
        const int x = (int)(item >> 16);
        lds_union(lab, off[yl] + (int)(item & 0xFFFFu), off[yl + 1] + (int)S.ord16[rb + w + x]);
    }
    __syncthreads();
    // sizes at the local roots; flatten
    for (int i = t; i < T; i += SPK_TPB) {
        const int yl = spk_row_of(off, R, i);
        const int len = (int)(S.runlist[ibase + (size_t)(y0 + yl) * w + (i - off[yl])] >> 16);
        const int r = lds_find(lab, i);
        atomicAdd(&lsize[r], len);
        if (r != i) __hip_atomic_store(&lab[i], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // roots do not move any more
    }
    __syncthreads();
    for (int i = t; i < T; i += SPK_TPB) {
        const int yl = spk_row_of(off, R, i);
        const size_t node = ibase + (size_t)(y0 + yl) * w + (i - off[yl]);
        const int r = lab[i];
        const int yr = spk_row_of(off, R, r);        // local index -> (row, ordinal)
        S.parent[node] = (y0 + yr) * w + (r - off[yr]);
        S.csize[node] = (r == i) ? lsize[i] : 0;
    }
}

constexpr int SPK_CROSS_TPB = 256;     // four waves per boundary row: a row of a 4096-wide map has hundreds of joins, each a chain of dependent accesses
// One workgroup per STRIP: the joins across the strip's lower seam (last row of the strip -> first row of the next), and -- only when the strip was
// too dense for the LDS level (global_strip) -- the joins of its interior rows as well.  (One workgroup per image row, seven of eight of which
// found nothing to do, was 95 744 waves per 64 KITTI-size maps for 11 968 rows with work.)
__global__ void __launch_bounds__(SPK_CROSS_TPB) k_spk_cross(SpkScratch S, int w, int h, int nstrips)
{
    const int strip = blockIdx.x;
    const bool all_rows = S.global_strip[blockIdx.z * (size_t)nstrips + strip] != 0;
    const size_t ibase = blockIdx.z * (size_t)w * h;
    int* nd = S.parent + ibase;
    const int ylast = (strip + 1) * SPK_STRIP - 1;                          // joins between rows y and y+1
    for (int y = all_rows ? strip * SPK_STRIP : ylast; y <= ylast && y < h - 1; y++) {
        const size_t rb = ibase + (size_t)y * w;
        const int n = S.njoin[blockIdx.z * (size_t)h + y];
        for (int k0 = 0; k0 < n; k0 += SPK_CROSS_TPB) {
            const int k = k0 + (int)threadIdx.x;
            int a = 0, b = 0;
            if (k < n) {
                const u32 item = S.joinlist[rb + k];
                const int x = (int)(item >> 16);
                // the strip-local roots of the two runs (or, concurrently relinked, an ancestor: all a union needs); neighbouring joins mostly
                // connect the same two large components: one union per run of equal pairs
                a = __hip_atomic_load(&nd[y * w + (int)(item & 0xFFFFu)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b = __hip_atomic_load(&nd[(y + 1) * w + (int)S.ord16[rb + w + x]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (first_of_key_run(k < n, ((u64)(u32)a << 32) | (u64)(u32)b)) uf_union(nd, a, b);
        }
    }
}

__global__ void __launch_bounds__(64) k_spk_sizes(SpkScratch S, int w, int h)
{
    const int y = blockIdx.x;
    const size_t ibase = blockIdx.z * (size_t)w * h;
    const int n = S.nrun[blockIdx.z * (size_t)h + y];
    int* nd = S.parent + ibase;
    for (int k = threadIdx.x; k < n; k += 64) {
        const int i = y * w + k;
        const int sz = S.csize[ibase + i];
        if (sz == 0) continue;                          // not a strip-local root
        int root = nd[i];
        if (root == i) {                                // still a root: its size stays where it is
            S.joinlist[ibase + i] = 0x7FFFFFFFu;        // (row-tile split: k_spk_border_link takes the minimum border pixel of the component here;
            continue;                                   //  the join lists are dead once k_spk_cross has run)
        }
        int q = nd[root];
        while (q != root) { root = q; q = nd[root]; }   // k_spk_cross is complete: plain loads are coherent
        nd[i] = root;                                   // compression: local roots now point straight at the global root
        atomicAdd(&S.csize[ibase + root], sz);
    }
}

__global__ void __launch_bounds__(64) k_spk_apply(int16_t* __restrict__ disp, size_t dpitch, SpkScratch S, int w, int h, int new_val, int max_size,
                                                   size_t dbs)
{
    const int y = blockIdx.x, lane = threadIdx.x;
    const size_t ibase = blockIdx.z * (size_t)w * h;
    const int n = S.nrun[blockIdx.z * (size_t)h + y];
    int16_t* drow = (int16_t*)((uint8_t*)disp + blockIdx.z * dbs) + (size_t)y * dpitch;
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + lane;
        u32 item = 0;
        bool kill = false;
        if (k < n) {
            item = S.runlist[ibase + (size_t)y * w + k];
            const int r1 = S.parent[ibase + (size_t)y * w + k];   // strip-local root (or already the global one)
            const int r2 = S.parent[ibase + r1];                   // k_spk_sizes left the global root in every local root
            kill = S.csize[ibase + r2] <= max_size;
        }
        // the wave erases the doomed runs together (a doomed run is at most maxSpeckleSize pixels long)
        unsigned long long m = __ballot(kill);
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const u32 it = (u32)__shfl((int)item, src);
            const int s0 = (int)(it & 0xFFFFu), len = (int)(it >> 16);
            for (int i = lane; i < len; i += 64) drow[s0 + i] = (int16_t)new_val;
        }
    }
}

// Labels of one image row after k_spk_sizes: per pixel the GLOBAL root node of its component (-1 for invalid pixels) and the size
// stored there.  Used by the row-tile split (rowsplit.py): components that touch a tile's first / last row continue in the
// neighbouring rank's tile; the ranks exchange these rows, merge the touching components and patch the sizes of their own roots
// (k_spk_set_sizes) before k_spk_apply.
__global__ void __launch_bounds__(256) k_spk_row_labels(SpkScratch S, int w, int h, int y, int* __restrict__ root_out, int* __restrict__ size_out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= w) return;
    const size_t ibase = blockIdx.z * (size_t)w * h;
    const unsigned ord = S.ord16[ibase + (size_t)y * w + x];
    int root = -1, size = 0;
    if (ord != 0xFFFFu) {
        const int r1 = S.parent[ibase + (size_t)y * w + ord];
        root = S.parent[ibase + r1];
        size = S.csize[ibase + root];
    }
    root_out[blockIdx.z * (size_t)w + x] = root;
    size_out[blockIdx.z * (size_t)w + x] = size;
}

__global__ void __launch_bounds__(256) k_spk_set_sizes(SpkScratch S, const int* __restrict__ roots, const int* __restrict__ sizes, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) S.csize[roots[i]] = sizes[i];
}

// ---- row-tile split (BASELINE configs[4]): components that cross tile borders, merged ON THE DEVICE --------------------------------
// A frame cut into N row tiles is labelled tile by tile (gipk_speckle_label on every tile's own rows).  What the tiles must agree on is
// the size of the components that continue across a border.  Each tile publishes a BORDER RECORD of its first and last row: for the 2w
// border pixels p (p < w: first row, else last row)
//     val  int16[2w] @ byte 0     the disparity
//     par  int32[2w] @ byte 4w    union-find parent, as an int32 index into the buffer that holds the records of ALL tiles one after the
//                                 other (record stride S ints): the node of pixel p of tile t is  t*S + w + p.  Initially the node of the
//                                 component's CANONICAL border pixel (the smallest p of that component in this tile); an invalid pixel
//                                 points at itself
//     size int32[2w] @ byte 12w   tile-local size of the component at its canonical pixel, 0 elsewhere  (node + 2w)
// 20 bytes per image column.  Once the records of all tiles sit side by side on a device (hipMemcpyPeerAsync / an all-gather) they ARE a
// union-find forest over the 2wN border pixels with every tile's components pre-linked: k_spk_border_union unites the pixels that face
// each other across a border when cv::filterSpeckles' rule connects them, k_spk_border_accum adds the sizes of the absorbed components
// to their roots, k_spk_border_patch overwrites the sizes of the tile's OWN border components with the total -- k_spk_apply then erases
// exactly what the filter on the full frame erases.  No host round trip.  All kernels are grids of one thread per border pixel: a
// first version with ONE workgroup per step (dependent global accesses in a loop) took 78 + 93 us per tile and frame.
constexpr int BRD_TPB = 256;
__host__ __device__ inline size_t spk_border_record_bytes(int w) { return ((size_t)20 * w + 255) / 256 * 256; }

// canon[root] (the join-list slot of the root node; k_spk_sizes has set it to INT_MAX) <- smallest border pixel of the component
__global__ void __launch_bounds__(BRD_TPB) k_spk_border_link(SpkScratch S, int w, int h, int* __restrict__ roots)
{
    const int p = blockIdx.x * BRD_TPB + threadIdx.x;
    int root = -1;
    if (p < 2 * w) {
        const int row = p < w ? 0 : h - 1, x = p < w ? p : p - w;
        const unsigned ord = S.ord16[(size_t)row * w + x];
        if (ord != 0xFFFFu) {
            const int r1 = S.parent[(size_t)row * w + ord];
            root = S.parent[r1];            // k_spk_sizes left the global root in every strip-local root
        }
        roots[p] = root;
    }
    // the first lane of a run of equal roots holds the smallest p of the run
    if (first_of_key_run(root >= 0, (u64)(u32)root)) __hip_atomic_fetch_min((int*)S.joinlist + root, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(BRD_TPB) k_spk_border_record(const int16_t* __restrict__ disp, size_t dpitch, SpkScratch S, int w, int h, int tile, uint8_t* __restrict__ rec,
                                                                const int* __restrict__ roots)
{
    const int p = blockIdx.x * BRD_TPB + threadIdx.x;
    if (p >= 2 * w) return;
    const int row = p < w ? 0 : h - 1, x = p < w ? p : p - w;
    const int base = tile * (int)(spk_border_record_bytes(w) / 4) + w;      // node of this tile's pixel 0
    const int root = roots[p];
    int canon = p, size = 0;
    if (root >= 0) {
        canon = ((const int*)S.joinlist)[root];
        if (canon == p) size = S.csize[root];
    }
    ((int16_t*)rec)[p] = disp[(size_t)row * dpitch + x];
    ((int*)(rec + (size_t)4 * w))[p] = base + canon;
    ((int*)(rec + (size_t)12 * w))[p] = size;
}

// nodes = int32 view of the gathered records; S = record stride in ints
__global__ void __launch_bounds__(BRD_TPB) k_spk_border_union(uint8_t* recs, int S, int n_tiles, int w, int new_val, int max_diff)
{
    const int i = blockIdx.x * BRD_TPB + threadIdx.x;
    bool on = false;
    int a = 0, c = 0;
    if (i < (n_tiles - 1) * w) {
        const int b = i / w, x = i - b * w;
        const int vb = ((const int16_t*)(recs + (size_t)b * S * 4))[w + x], vt = ((const int16_t*)(recs + (size_t)(b + 1) * S * 4))[x];
        if (spk_conn(vb, vt, new_val, max_diff)) {
            // the two pixels' parents: the canonical pixels of their components when this kernel starts (possibly relinked by now: still
            // an ancestor, which is all a union needs) -- equal for every pair that joins the same two components
            on = true;
            a = __hip_atomic_load((int*)recs + b * S + w + (w + x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c = __hip_atomic_load((int*)recs + (b + 1) * S + w + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (first_of_key_run(on, ((u64)(u32)a << 32) | (u64)(u32)c)) uf_union((int*)recs, a, c);
}

// the records of all tiles -> this tile's working copy, ONE launch: src[t] may be peer memory (hipDeviceEnablePeerAccess)
struct BorderSrc { const uint8_t* p[16]; };
__global__ void __launch_bounds__(BRD_TPB) k_spk_border_gather(BorderSrc src, uint8_t* __restrict__ dst, int S4 /* record stride in 16-byte units */, int n_used4, int n_tiles)
{
    const int i = blockIdx.x * BRD_TPB + threadIdx.x;
    if (i >= n_used4 * n_tiles) return;
    const int t = i / n_used4, k = i - t * n_used4;
    ((uint4*)dst)[(size_t)t * S4 + k] = ((const uint4*)src.p[t])[k];
}

__global__ void __launch_bounds__(BRD_TPB) k_spk_border_accum(uint8_t* recs, int S, int n_tiles, int w)
{
    const int i = blockIdx.x * BRD_TPB + threadIdx.x, n2 = 2 * w;
    if (i >= n_tiles * n2) return;
    const int t = i / n2, p = i - t * n2, g = t * S + w + p;
    int* nd = (int*)recs;
    const int sz = nd[g + n2];
    if (sz == 0) return;                       // not a canonical pixel
    const int R = uf_find(nd, g);
    if (R != g) atomicAdd(&nd[R + n2], sz);    // an absorbed component: nobody adds to ITS size, so the plain read above is safe
}

__global__ void __launch_bounds__(BRD_TPB) k_spk_border_patch(uint8_t* recs, int S, int me, int w, SpkScratch Sc, const int* __restrict__ roots)
{
    const int p = blockIdx.x * BRD_TPB + threadIdx.x, n2 = 2 * w;
    if (p >= n2) return;
    const int root = roots[p];
    if (root < 0 || ((const int*)Sc.joinlist)[root] != p) return;      // invalid, or not the component's canonical pixel
    int* nd = (int*)recs;
    Sc.csize[root] = nd[uf_find(nd, me * S + w + p) + n2];
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
                               size_t dbs, size_t cbs, size_t fbs, size_t obs, int y_org)
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
    const int yf = y + y_org;      // row of the full frame (row-tile split: the tile holds rows [y_org, y_org + h))
    double qx = q[1] * yf + q[3] + q[0] * x, qy = q[5] * yf + q[7] + q[4] * x;
    double qz = q[9] * yf + q[11] + q[8] * x, qw = q[13] * yf + q[15] + q[12] * x;
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

// The same for the Q of a rectified stereo pair (image_geometry::StereoCameraModel::updateQ): only Q00 Q03 Q11 Q13 Q23 Q32 Q33 are
// non-zero.  Dropping the zero terms changes no bit (x * +0.0 and + +0.0 are exact; the launcher also checks the zeros are
// positive zeros) but more than halves the double-precision work.  This kernel runs next to the VALU-bound matcher of another batch, so its instruction count matters.
constexpr int OUT_ROWS = 1;   // 4 rows per thread (64x4 blocks) halved the store bandwidth: a workgroup must write long contiguous runs
// Everything the kernel derives from the disparity VALUE alone, tabulated once per (model, parameter) change: the float disparity, the
// reciprocal of W = Q33 + Q32 d (an FP64 division per pixel otherwise) and Z.  One 16-byte load replaces them; the entries are computed
// with the very expressions of the kernel, so the outputs are bit-identical (tests/test_gpu_kernels.py).  Index = d16 - lo.
struct __attribute__((aligned(16))) WEntry { double iw; float z; float df; };
__global__ void k_w_table(WEntry* __restrict__ tab, int n, int lo, QMat Q, float alpha, float beta, int filtered)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d16 = lo + i;
    float df = (float)d16 * alpha;
    df = df + beta;
    const double d = df;
    const double iW = 1. / (Q.q[15] + Q.q[14] * d);
    double Z = Q.q[11] * iW;
    if (d16 == filtered) Z = 10000.;
    // GPUSenderPc2::fillInData (GpuSenderPc2.cpp:47-57): a point whose Z is MISSING_Z or infinite becomes NaN NaN NaN.  Folded into the entry:
    // x * NaN and y * NaN are NaN, so the consumers need no test of their own (store_pc2_record's test then passes NaN through unchanged)
    const float zf = (float)Z;
    const bool valid = (zf != 10000.f) && !isinf(zf);
    tab[i].iw = valid ? iW : __builtin_nan("");
    tab[i].z = valid ? zf : __builtin_nanf("");
    tab[i].df = df;
}

// The cloud kernel of the batch chain: every disparity comes from this library's own matcher + speckle filter, i.e. lies inside the table
// (FILTERED .. (minD + ndisp) * 16; the index is clamped, so foreign values cannot read outside it but get the clamped entry -- callers that may
// hold foreign values use k_disp_outputs_sq, which falls back to the arithmetic).  One image row per workgroup row: every base address is
// wave-uniform (scalar registers), the lane adds a 32-bit offset; 27 vector instructions per pixel against 89 of the general kernel.
// Measured in the pipelined bench (same box, alternating runs, profiles/r03_outputs_tab_ab.txt): every workload whose step is not bound by the
// matcher chain alone gains -- euroc +7.3 %, 1242x375 at ndisp 64 / 96 +11 %, ndisp 112 +2.4 %, refdefault +2.5 %, hd1080 +1.8 %, uhd4k +1.4 % --
// and the two shapes at the cross-over lose a little (ndisp 128: -0.35 %, ndisp 144: -0.2 %).
template <int CN>
__global__ void __launch_bounds__(TPB) k_disp_outputs_tab(const int16_t* __restrict__ disp, size_t dpitch, const uint8_t* __restrict__ col, size_t cpitch,
                                                           float* __restrict__ f32out, uint8_t* __restrict__ pc, int w, double q0, double q3, double q5, double q7,
                                                           size_t dbs, size_t cbs, size_t fbs, size_t obs, int y_org, const WEntry* __restrict__ tab, int tab_lo,
                                                           int tab_n)
{
    const unsigned x = blockIdx.x * TPB + threadIdx.x;
    if (x >= (unsigned)w) return;
    const unsigned y = blockIdx.y, z = blockIdx.z;
    const int16_t* drow = (const int16_t*)((const uint8_t*)disp + (size_t)z * dbs) + (size_t)y * dpitch;
    const int d16 = drow[x];
    // the colour load is independent of the table lookup: issue it before the dependent load
    u32 c0 = 0, c1 = 0, c2 = 0;
    if (pc && col) {
        const uint8_t* crow = col + (size_t)z * cbs + (size_t)y * cpitch;
        if (CN == 1) c0 = crow[x];
        else { c0 = crow[3u * x]; c1 = crow[3u * x + 1]; c2 = crow[3u * x + 2]; }
    }
    int ti = d16 - tab_lo;
    ti = ti < 0 ? 0 : (ti >= tab_n ? tab_n - 1 : ti);
    const WEntry e = tab[(unsigned)ti];
    if (f32out) {
        float* frow = (float*)((uint8_t*)f32out + (size_t)z * fbs) + (size_t)y * w;
        frow[x] = e.df;
    }
    if (!pc) return;
    const u32 c = CN == 1 ? c0 * 0x010101u : (c0 | (c1 << 8) | (c2 << 16));
    const double qx = q3 + q0 * (int)x;
    const double qy = q5 * (int)(y + y_org) + q7;
    uint8_t* prow = pc + (size_t)z * obs + (size_t)y * w * 32;
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store((f4){(float)(qx * e.iw), (float)(qy * e.iw), e.z, 0.f}, (f4*)(prow + 32u * x));
    __builtin_nontemporal_store((u4){c, 0u, 0u, 0u}, (u4*)(prow + 32u * x) + 1);
}

template <int CN>
__global__ void k_disp_outputs_sq(const int16_t* __restrict__ disp, size_t dpitch, const uint8_t* __restrict__ col, size_t cpitch,
                                  float* __restrict__ f32out, uint8_t* __restrict__ pc, int w, int h, QMat Q, float alpha, float beta, int filtered,
                                  size_t dbs, size_t cbs, size_t fbs, size_t obs, int y_org, const WEntry* __restrict__ tab, int tab_lo, int tab_n)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y0 = (blockIdx.y * blockDim.y + threadIdx.y) * OUT_ROWS;
    if (x >= w || y0 >= h) return;
    const double* q = Q.q;
    const int16_t* dp0 = (const int16_t*)((const uint8_t*)disp + blockIdx.z * dbs) + (size_t)y0 * dpitch + x;
    int d16[OUT_ROWS];
#pragma unroll
    for (int j = 0; j < OUT_ROWS; j++) d16[j] = (y0 + j < h) ? (int)dp0[(size_t)j * dpitch] : filtered;
    u32 c[OUT_ROWS];
#pragma unroll
    for (int j = 0; j < OUT_ROWS; j++) {
        c[j] = 0;
        if (col && pc && y0 + j < h) {
            const uint8_t* cp = col + blockIdx.z * cbs + (size_t)(y0 + j) * cpitch + (size_t)CN * x;
            if (CN == 1) c[j] = (u32)cp[0] * 0x010101u;
            else c[j] = (u32)cp[0] | ((u32)cp[1] << 8) | ((u32)cp[2] << 16);
        }
    }
    const double qx = q[3] + q[0] * x;
#pragma unroll
    for (int j = 0; j < OUT_ROWS; j++) {
        const int y = y0 + j;
        if (y >= h) break;
        const double qy = q[5] * (y + y_org) + q[7];
        const unsigned ti = (unsigned)(d16[j] - tab_lo);
        if (tab && ti < (unsigned)tab_n) {
            const WEntry e = tab[ti];
            if (f32out) ((float*)((uint8_t*)f32out + blockIdx.z * fbs))[(size_t)y * w + x] = e.df;
            if (!pc) continue;
            store_pc2_record(pc + blockIdx.z * obs + ((size_t)y * w + x) * 32, (float)(qx * e.iw), (float)(qy * e.iw), e.z, c[j]);
            continue;
        }
        // No table, or a disparity outside its range (never a matcher output).  The compiler hoists this arithmetic above the branch and runs it
        // for every pixel NEXT TO the table path (89 instead of 72 wave-instructions per pixel, profiles/r03_pmc_sq_all_kernels.txt) -- and the
        // pipelined step is 0.8 % FASTER with the table than without, and 1.2 % SLOWER when the hoisting is prevented (60 instructions per pixel;
        // profiles/r03_outputs_table_ab.txt): the dependent table load then sits alone on the critical path of every wave.  Left as the compiler
        // schedules it.
        const int dd = d16[j];
        float df = (float)dd * alpha;
        df = df + beta;
        if (f32out) ((float*)((uint8_t*)f32out + blockIdx.z * fbs))[(size_t)y * w + x] = df;
        if (!pc) continue;
        const double d = df;
        const double iW = 1. / (q[15] + q[14] * d);
        const double X = qx * iW, Y = qy * iW;
        double Z = q[11] * iW;
        if (dd == filtered) Z = 10000.;
        store_pc2_record(pc + blockIdx.z * obs + ((size_t)y * w + x) * 32, (float)X, (float)Y, (float)Z, c[j]);
    }
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
    return gip_bm_finalize_impl(disp, disp_pitch, cost, cost_pitch, w, h, p, batch, disp_bstride, cost_bstride, stream, false);
}

}  // extern "C"

template <class CT>
static int finalize_launch(int16_t* disp, size_t disp_pitch, const CT* cost, size_t cost_pitch, int w, int h, const gip_bm_params* p, int batch, size_t disp_bstride,
                           size_t cost_bstride, gip_stream stream)
{
    const GipBmGeom g = gip_bm_geom(w, h, p);
    size_t sh = p->disp12_max_diff >= 0 ? (size_t)w * 8 : 16;
    if (sh > 64 * 1024) GIP_HIP_CHECK(hipFuncSetAttribute((const void*)k_bm_finalize<CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    int rx = g.roi_x, ry = g.roi_y, rw = g.roi_w, rh = g.roi_h;
    if (g.degenerate) { rx = ry = rw = rh = 0; }
#define GIP_FINALIZE_ARGS disp, disp_pitch / 2, cost, cost ? cost_pitch / sizeof(CT) : 0, w, h, p->min_disparity, p->num_disparities, g.degenerate ? -1 : p->disp12_max_diff, rx, ry, \
                          rw, rh, g.lofs, g.lofs + g.width1, disp_bstride, cost_bstride
    const dim3 grid(h, 1, batch);
    // one-dword keys for wide rows (where the cost is a uint16 and d - minD * 16 fits 12 bits)
#ifndef K32_MIN_W
#define K32_MIN_W 1536     /* measured: 1242-wide rows lose with the one-dword form, 1920-wide gain 1 %, 4096-wide 13 % */
#endif
    const bool k32 = sizeof(CT) == 2 && p->num_disparities <= 256 && p->disp12_max_diff >= 0 && w > K32_MIN_W;
    const size_t sh32 = (size_t)w * 4;
#define GIP_FINALIZE_R(NI)                                                                                                                       \
    do {                                                                                                                                         \
        if (k32) hipLaunchKernelGGL((k_bm_finalize_r<NI, CT, sizeof(CT) == 2>), grid, dim3(TPB), sh32, (hipStream_t)stream, GIP_FINALIZE_ARGS);  \
        else hipLaunchKernelGGL((k_bm_finalize_r<NI, CT, false>), grid, dim3(TPB), sh, (hipStream_t)stream, GIP_FINALIZE_ARGS);                   \
    } while (0)
    if (w <= 2 * TPB) GIP_FINALIZE_R(2);
    else if (w <= 5 * TPB) GIP_FINALIZE_R(5);
    else if (w <= 8 * TPB) GIP_FINALIZE_R(8);
    else if (w <= 16 * TPB) GIP_FINALIZE_R(16);
    else hipLaunchKernelGGL(k_bm_finalize<CT>, grid, dim3(TPB), sh, (hipStream_t)stream, GIP_FINALIZE_ARGS);
#undef GIP_FINALIZE_R
#undef GIP_FINALIZE_ARGS
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gip_bm_finalize_impl(int16_t* disp, size_t disp_pitch, const void* cost, size_t cost_pitch, int w, int h, const gip_bm_params* p, int batch, size_t disp_bstride,
                         size_t cost_bstride, gip_stream stream, bool cost_u16)
{
    GIPK_ARGCHECK(disp && p && w > 0 && h > 0 && batch > 0 && disp_pitch % 2 == 0);
    GIPK_ARGCHECK(p->disp12_max_diff < 0 || (cost && cost_pitch % 4 == 0));
    if (w > 16384) return GIP_ERR_UNSUPPORTED;
    if (cost_u16) return finalize_launch<uint16_t>(disp, disp_pitch, (const uint16_t*)cost, cost_pitch, w, h, p, batch, disp_bstride, cost_bstride, stream);
    return finalize_launch<int32_t>(disp, disp_pitch, (const int32_t*)cost, cost_pitch, w, h, p, batch, disp_bstride, cost_bstride, stream);
}

extern "C" {

int gipk_disparity_to_f32(const int16_t* disp, size_t disp_pitch, float* dst, size_t dst_pitch, int w, int h, double cx_l, double cx_r, int batch,
                          size_t disp_bstride, size_t dst_bstride, gip_stream stream)
{
    GIPK_ARGCHECK(disp && dst && w > 0 && h > 0 && batch > 0 && disp_pitch % 2 == 0 && dst_pitch % 4 == 0);
    hipLaunchKernelGGL(k_disp_to_f32, grid2d(w, h, batch), dim3(TPB), 0, (hipStream_t)stream, disp, disp_pitch / 2, dst, dst_pitch / 4, w, h,
                       (float)(1.0 / 16), (float)(-(cx_l - cx_r)), disp_bstride, dst_bstride);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

size_t gipk_speckle_scratch_bytes(int w, int h, int batch) { return (size_t)w * h * batch * 18 + (size_t)h * batch * 12 + 64; }

static int spk_carve(void* scratch, size_t scratch_bytes, int w, int h, int batch, SpkScratch* S)
{
    if (!scratch || scratch_bytes < gipk_speckle_scratch_bytes(w, h, batch) || (size_t)w * h >= (1u << 31)) return GIP_ERR_INVALID_ARG;
    if (w > 32768) return GIP_ERR_UNSUPPORTED;
    const size_t n = (size_t)w * h * batch;
    S->parent = (int*)gip_align_up((size_t)(uintptr_t)scratch, 16);
    S->csize = S->parent + n;
    S->runlist = (u32*)(S->csize + n);
    S->joinlist = S->runlist + n;
    S->nrun = (int*)(S->joinlist + n);
    S->njoin = S->nrun + (size_t)h * batch;
    S->global_strip = S->njoin + (size_t)h * batch;    // nstrips * batch <= h * batch entries
    S->ord16 = (unsigned short*)(S->global_strip + (size_t)h * batch);
    return GIP_OK;
}

// phase 1 of cv::filterSpeckles: label the components and leave their sizes at the global roots (no pixel is changed)
int gipk_speckle_label(const int16_t* disp, size_t disp_pitch, int w, int h, int new_val, int max_diff, void* scratch, size_t scratch_bytes, int batch,
                       size_t disp_bstride, gip_stream stream)
{
    GIPK_ARGCHECK(disp && w > 0 && h > 0 && batch > 0 && disp_pitch % 2 == 0);
    SpkScratch S;
    int rc = spk_carve(scratch, scratch_bytes, w, h, batch, &S);
    if (rc != GIP_OK) return rc;
    const int nstrips = (h + SPK_STRIP - 1) / SPK_STRIP;
    hipStream_t st = (hipStream_t)stream;
    const bool pairs_ok = (w % 2 == 0) && (disp_pitch % 4 == 0) && ((uintptr_t)disp % 4 == 0) && (disp_bstride % 4 == 0) && ((uintptr_t)S.ord16 % 4 == 0);
    if (pairs_ok && w >= SPK_SCAN4_MIN_W) hipLaunchKernelGGL(k_spk_scan4, dim3(h, 1, batch), dim3(64), 0, st, disp, disp_pitch / 2, S, w, h, new_val, max_diff, disp_bstride);
    else if (pairs_ok) hipLaunchKernelGGL(k_spk_scan2, dim3(h, 1, batch), dim3(64), 0, st, disp, disp_pitch / 2, S, w, h, new_val, max_diff, disp_bstride);
    else hipLaunchKernelGGL(k_spk_scan, dim3(h, 1, batch), dim3(64), 0, st, disp, disp_pitch / 2, S, w, h, new_val, max_diff, disp_bstride);
    hipLaunchKernelGGL(k_spk_strip, dim3(nstrips, 1, batch), dim3(SPK_TPB), (size_t)SPK_CAP * 8, st, S, w, h, nstrips);
    if (h > 1) hipLaunchKernelGGL(k_spk_cross, dim3(nstrips, 1, batch), dim3(SPK_CROSS_TPB), 0, st, S, w, h, nstrips);
    hipLaunchKernelGGL(k_spk_sizes, dim3(h, 1, batch), dim3(64), 0, st, S, w, h);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

// phase 2: erase the components whose size (as stored at their roots) is <= max_speckle_size
int gipk_speckle_apply(int16_t* disp, size_t disp_pitch, int w, int h, int new_val, int max_speckle_size, void* scratch, size_t scratch_bytes, int batch,
                       size_t disp_bstride, gip_stream stream)
{
    GIPK_ARGCHECK(disp && w > 0 && h > 0 && batch > 0 && disp_pitch % 2 == 0);
    SpkScratch S;
    int rc = spk_carve(scratch, scratch_bytes, w, h, batch, &S);
    if (rc != GIP_OK) return rc;
    hipLaunchKernelGGL(k_spk_apply, dim3(h, 1, batch), dim3(64), 0, (hipStream_t)stream, disp, disp_pitch / 2, S, w, h, new_val, max_speckle_size, disp_bstride);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_speckle_row_labels(void* scratch, size_t scratch_bytes, int w, int h, int batch, int row, int32_t* root_out, int32_t* size_out, gip_stream stream)
{
    GIPK_ARGCHECK(root_out && size_out && w > 0 && h > 0 && batch > 0 && row >= 0 && row < h);
    SpkScratch S;
    int rc = spk_carve(scratch, scratch_bytes, w, h, batch, &S);
    if (rc != GIP_OK) return rc;
    hipLaunchKernelGGL(k_spk_row_labels, dim3(gip_div_up(w, 256), 1, batch), dim3(256), 0, (hipStream_t)stream, S, w, h, row, root_out, size_out);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_speckle_set_sizes(void* scratch, size_t scratch_bytes, int w, int h, int batch, const int32_t* roots, const int32_t* sizes, int n, gip_stream stream)
{
    GIPK_ARGCHECK(w > 0 && h > 0 && batch == 1 && n >= 0 && (n == 0 || (roots && sizes)));
    if (n == 0) return GIP_OK;
    SpkScratch S;
    int rc = spk_carve(scratch, scratch_bytes, w, h, batch, &S);
    if (rc != GIP_OK) return rc;
    hipLaunchKernelGGL(k_spk_set_sizes, dim3(gip_div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, S, roots, sizes, n);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

size_t gipk_speckle_border_record_bytes(int w) { return w > 0 ? spk_border_record_bytes(w) : 0; }

int gipk_speckle_border_record(const int16_t* disp, size_t disp_pitch, int w, int h, int tile, void* scratch, size_t scratch_bytes, void* record, int32_t* border_roots,
                               gip_stream stream)
{
    GIPK_ARGCHECK(disp && record && border_roots && w > 0 && h > 0 && tile >= 0 && disp_pitch % 2 == 0 && ((uintptr_t)record & 3) == 0);
    GIPK_ARGCHECK((size_t)(tile + 1) * spk_border_record_bytes(w) < ((size_t)1 << 31));      // node indices are int32
    SpkScratch S;
    int rc = spk_carve(scratch, scratch_bytes, w, h, 1, &S);
    if (rc != GIP_OK) return rc;
    const dim3 grid(gip_div_up(2 * w, BRD_TPB));
    hipLaunchKernelGGL(k_spk_border_link, grid, dim3(BRD_TPB), 0, (hipStream_t)stream, S, w, h, border_roots);
    hipLaunchKernelGGL(k_spk_border_record, grid, dim3(BRD_TPB), 0, (hipStream_t)stream, disp, disp_pitch / 2, S, w, h, tile, (uint8_t*)record, border_roots);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_speckle_border_gather(const void* const* tile_records, int n_tiles, int w, void* records, gip_stream stream)
{
    GIPK_ARGCHECK(tile_records && records && n_tiles > 0 && n_tiles <= 16 && w > 0 && ((uintptr_t)records & 15) == 0);
    BorderSrc src{};
    for (int t = 0; t < n_tiles; t++) {
        GIPK_ARGCHECK(tile_records[t] && ((uintptr_t)tile_records[t] & 15) == 0);
        src.p[t] = (const uint8_t*)tile_records[t];
    }
    const int used4 = (int)(((size_t)20 * w + 15) / 16), stride4 = (int)(spk_border_record_bytes(w) / 16);
    hipLaunchKernelGGL(k_spk_border_gather, dim3(gip_div_up(used4 * n_tiles, BRD_TPB)), dim3(BRD_TPB), 0, (hipStream_t)stream, src, (uint8_t*)records, stride4, used4, n_tiles);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_speckle_border_merge(void* records, int n_tiles, int tile, int w, int h, int new_val, int max_diff, void* scratch, size_t scratch_bytes,
                              const int32_t* border_roots, gip_stream stream)
{
    GIPK_ARGCHECK(records && border_roots && w > 0 && h > 0 && n_tiles > 0 && tile >= 0 && tile < n_tiles && ((uintptr_t)records & 3) == 0);
    GIPK_ARGCHECK((size_t)n_tiles * spk_border_record_bytes(w) < ((size_t)1 << 31));
    if (n_tiles == 1) return GIP_OK;       // nothing crosses a border
    SpkScratch S;
    int rc = spk_carve(scratch, scratch_bytes, w, h, 1, &S);
    if (rc != GIP_OK) return rc;
    const int stride = (int)(spk_border_record_bytes(w) / 4);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_spk_border_union, dim3(gip_div_up((n_tiles - 1) * w, BRD_TPB)), dim3(BRD_TPB), 0, st, (uint8_t*)records, stride, n_tiles, w, new_val, max_diff);
    hipLaunchKernelGGL(k_spk_border_accum, dim3(gip_div_up(n_tiles * 2 * w, BRD_TPB)), dim3(BRD_TPB), 0, st, (uint8_t*)records, stride, n_tiles, w);
    hipLaunchKernelGGL(k_spk_border_patch, dim3(gip_div_up(2 * w, BRD_TPB)), dim3(BRD_TPB), 0, st, (uint8_t*)records, stride, tile, w, S, border_roots);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

int gipk_filter_speckles(int16_t* disp, size_t disp_pitch, int w, int h, int new_val, int max_speckle_size, int max_diff, void* scratch,
                         size_t scratch_bytes, int batch, size_t disp_bstride, gip_stream stream)
{
    int rc = gipk_speckle_label(disp, disp_pitch, w, h, new_val, max_diff, scratch, scratch_bytes, batch, disp_bstride, stream);
    if (rc != GIP_OK) return rc;
    return gipk_speckle_apply(disp, disp_pitch, w, h, new_val, max_speckle_size, scratch, scratch_bytes, batch, disp_bstride, stream);
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
    return gipk_disparity_outputs_rows(disp, disp_pitch, color, color_pitch, color_cn, f32_out, pc_out, w, h, 0, Q, cx_l, cx_r, min_disparity, batch, disp_bstride,
                                       color_bstride, stream);
}

int gipk_disparity_outputs_rows(const int16_t* disp, size_t disp_pitch, const uint8_t* color, size_t color_pitch, int color_cn, float* f32_out,
                                uint8_t* pc_out, int w, int h, int y_origin, const double Q[16], double cx_l, double cx_r, int min_disparity, int batch,
                                size_t disp_bstride, size_t color_bstride, gip_stream stream)
{
    return gip_disparity_outputs_impl(disp, disp_pitch, color, color_pitch, color_cn, f32_out, pc_out, w, h, y_origin, Q, cx_l, cx_r, min_disparity, batch, disp_bstride,
                                      color_bstride, stream, nullptr, 0);
}

}  // extern "C"

size_t gip_w_table_bytes(int ndisp) { return ndisp > 0 ? (size_t)((ndisp + 1) * 16 + 1) * sizeof(WEntry) : 0; }

int gip_w_table_build(void* table, const double Q[16], double cx_l, double cx_r, int min_disparity, int ndisp, gip_stream stream)
{
    GIPK_ARGCHECK(table && Q && ndisp > 0 && ((uintptr_t)table & 15) == 0);
    QMat q;
    for (int i = 0; i < 16; i++) q.q[i] = Q[i];
    const int n = (ndisp + 1) * 16 + 1, lo = (min_disparity - 1) * 16;
    hipLaunchKernelGGL(k_w_table, dim3(gip_div_up(n, TPB)), dim3(TPB), 0, (hipStream_t)stream, (WEntry*)table, n, lo, q, (float)(1.0 / 16), (float)(-(cx_l - cx_r)), lo);
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

// w_table (optional): gip_w_table_build's output for the same Q, cx, min_disparity and `table_ndisp` disparities -- used by the kernel for the
// sparse Q of a rectified pair; a disparity outside the table's range falls back to the arithmetic
int gip_disparity_outputs_impl(const int16_t* disp, size_t disp_pitch, const uint8_t* color, size_t color_pitch, int color_cn, float* f32_out, uint8_t* pc_out, int w, int h,
                               int y_origin, const double Q[16], double cx_l, double cx_r, int min_disparity, int batch, size_t disp_bstride, size_t color_bstride,
                               gip_stream stream, const void* w_table, int table_ndisp, bool own_disparity)
{
    GIPK_ARGCHECK(disp && Q && (f32_out || pc_out) && w > 0 && h > 0 && batch > 0 && disp_pitch % 2 == 0);
    GIPK_ARGCHECK(!pc_out || ((uintptr_t)pc_out & 15) == 0);
    GIPK_ARGCHECK(!color || color_cn == 1 || color_cn == 3);
    QMat q;
    for (int i = 0; i < 16; i++) q.q[i] = Q[i];
    const float alpha = (float)(1.0 / 16), beta = (float)(-(cx_l - cx_r));
    const size_t npx = (size_t)w * h;
    // the Q of a rectified stereo pair: every entry outside {0, 3, 5, 7, 11, 14, 15} is +0.0
    bool stereo_q = true;
    for (int i : {1, 2, 4, 6, 8, 9, 10, 12, 13}) stereo_q = stereo_q && Q[i] == 0.0 && !std::signbit(Q[i]);
#define GIP_OUT_ARGS disp, disp_pitch / 2, color, color_pitch, f32_out, pc_out, w, h, q, alpha, beta, (min_disparity - 1) * 16, disp_bstride, color_bstride, npx * 4, npx * 32, y_origin
#define GIP_TAB_ARGS (const WEntry*)w_table, (min_disparity - 1) * 16, w_table ? (table_ndisp + 1) * 16 + 1 : 0
    const dim3 g4(gip_div_up(w, TPB), gip_div_up(h, OUT_ROWS), batch), b4(TPB, 1);
    if (stereo_q && w_table && own_disparity && w <= (1 << 24) && h <= 65535 && batch <= 65535 && (!color || color_cn == 1 || color_cn == 3)) {
        const dim3 g(gip_div_up(w, TPB), h, batch);
#define GIP_TABK_ARGS disp, disp_pitch / 2, color, color_pitch, f32_out, pc_out, w, Q[0], Q[3], Q[5], Q[7], disp_bstride, color_bstride, npx * 4, npx * 32, y_origin, \
                      (const WEntry*)w_table, (min_disparity - 1) * 16, (table_ndisp + 1) * 16 + 1
        if (color && color_cn == 3) hipLaunchKernelGGL(k_disp_outputs_tab<3>, g, dim3(TPB), 0, (hipStream_t)stream, GIP_TABK_ARGS);
        else hipLaunchKernelGGL(k_disp_outputs_tab<1>, g, dim3(TPB), 0, (hipStream_t)stream, GIP_TABK_ARGS);
#undef GIP_TABK_ARGS
        GIP_LAUNCH_CHECK();
        return GIP_OK;
    }
    if (color && color_cn == 1) {
        if (stereo_q) hipLaunchKernelGGL(k_disp_outputs_sq<1>, g4, b4, 0, (hipStream_t)stream, GIP_OUT_ARGS, GIP_TAB_ARGS);
        else hipLaunchKernelGGL(k_disp_outputs<1>, grid2d(w, h, batch), dim3(TPB), 0, (hipStream_t)stream, GIP_OUT_ARGS);
    } else {
        if (stereo_q) hipLaunchKernelGGL(k_disp_outputs_sq<3>, g4, b4, 0, (hipStream_t)stream, GIP_OUT_ARGS, GIP_TAB_ARGS);
        else hipLaunchKernelGGL(k_disp_outputs<3>, grid2d(w, h, batch), dim3(TPB), 0, (hipStream_t)stream, GIP_OUT_ARGS);
    }
#undef GIP_OUT_ARGS
#undef GIP_TAB_ARGS
    GIP_LAUNCH_CHECK();
    return GIP_OK;
}

extern "C" {

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
    GIPK_ARGCHECK(left && right);   // (null inputs mean "planes already prefiltered" to the internal entry point only)
    return gip_stereobm_impl(left, right, pitch, w, h, p, disp, disp_pitch, scratch, scratch_bytes, batch, img_bstride, disp_bstride, stream, nullptr);
}

}  // extern "C"

void gip_stereobm_planes(void* scratch, int w, int h, int batch, uint8_t** left_f, uint8_t** right_f, size_t* pitch, size_t* plane)
{
    const size_t pp = bm_plane_pitch(w);
    uint8_t* base = (uint8_t*)gip_align_up((size_t)(uintptr_t)scratch, 256);
    *left_f = base;
    *right_f = base + pp * h * batch;
    *pitch = pp;
    *plane = pp * h;
}

int gip_stereobm_impl(const uint8_t* left, const uint8_t* right, size_t pitch, int w, int h, const gip_bm_params* p, int16_t* disp, size_t disp_pitch,
                      void* scratch, size_t scratch_bytes, int batch, size_t img_bstride, size_t disp_bstride, gip_stream stream, const BmLaunchCtx* ctx)
{
    int cost_u16 = 1;
    return gip_stereobm_phase_impl(left, right, pitch, w, h, p, disp, disp_pitch, scratch, scratch_bytes, batch, img_bstride, disp_bstride, stream, ctx, 0, &cost_u16);
}

// phase 0: everything; 1: prefilter + matcher only (*cost_u16 tells phase 2 the type of the min-SAD plane); 2: validateDisparity (+ the
// matcher's own speckle window) only.  gip_batch_run's FRONT / BACK halves are phases 1 and 2.
int gip_stereobm_phase_impl(const uint8_t* left, const uint8_t* right, size_t pitch, int w, int h, const gip_bm_params* p, int16_t* disp, size_t disp_pitch,
                            void* scratch, size_t scratch_bytes, int batch, size_t img_bstride, size_t disp_bstride, gip_stream stream, const BmLaunchCtx* ctx,
                            int phase, int* cost_u16_io)
{
    const bool prefiltered = !left && !right;
    GIPK_ARGCHECK((prefiltered || (left && right)) && disp && scratch && p && w > 0 && h > 0 && batch > 0);
    int rc = gip_validate_bm_params(p, w, h);
    if (rc != GIP_OK) return rc;
    if (scratch_bytes < gipk_stereobm_scratch_bytes(w, h, batch, p)) return GIP_ERR_INVALID_ARG;
    const size_t pp = bm_plane_pitch(w), plane = pp * h;
    uint8_t* base = (uint8_t*)gip_align_up((size_t)(uintptr_t)scratch, 256);
    uint8_t* lf = base;
    uint8_t* rf = lf + plane * batch;
    int32_t* cost = (int32_t*)gip_align_up((size_t)(uintptr_t)(rf + plane * batch), 256);
    void* spk = (void*)gip_align_up((size_t)(uintptr_t)((uint8_t*)cost + (size_t)w * h * batch * 4), 256);
    if (!prefiltered && phase != 2) {
        rc = gipk_prefilter(left, pitch, lf, pp, w, h, p->prefilter_type, p->prefilter_size, p->prefilter_cap, batch, img_bstride, plane, stream);
        if (rc != GIP_OK) return rc;
        rc = gipk_prefilter(right, pitch, rf, pp, w, h, p->prefilter_type, p->prefilter_size, p->prefilter_cap, batch, img_bstride, plane, stream);
        if (rc != GIP_OK) return rc;
    }
    const bool lr = p->disp12_max_diff >= 0;
    // min-SAD plane for the L/R check: uint16 when the column-sum matcher runs (its sums are < 65535), else int32; the byte pitch is
    // the same (rows of 4 * w bytes, half used by a uint16 plane), so the scratch layout does not depend on the choice
    int cost_u16 = *cost_u16_io;
    if (phase != 2) {
        rc = gip_bm_match_impl(lf, rf, pp, w, h, p, disp, disp_pitch, lr ? cost : nullptr, (size_t)w * 4, batch, plane, disp_bstride, (size_t)w * h * 4, stream, true, ctx, &cost_u16);
        if (rc != GIP_OK) return rc;
        *cost_u16_io = cost_u16;
        if (phase == 1) return GIP_OK;
    }
    rc = gip_bm_finalize_impl(disp, disp_pitch, lr ? cost : nullptr, (size_t)w * 4, w, h, p, batch, disp_bstride, (size_t)w * h * 4, stream, cost_u16 != 0);
    if (rc != GIP_OK) return rc;
    if (p->speckle_range >= 0 && p->speckle_window_size > 0)
        rc = gipk_filter_speckles(disp, disp_pitch, w, h, (p->min_disparity - 1) * 16, p->speckle_window_size, p->speckle_range, spk,
                                  gipk_speckle_scratch_bytes(w, h, batch), batch, disp_bstride, stream);
    return rc;
}
