// Micro-benchmark: issue rate of the packed-byte SAD family and packed-u16 VALU ops on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu tools/ubench_valu.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;

#define KERNEL64(NAME, ASM)                                                                                   \
__global__ void NAME(uint64_t* out, uint32_t seed) {                                                          \
    uint64_t acc[UNROLL];                                                                                     \
    uint64_t a = ((uint64_t)(threadIdx.x * 2654435761u + seed) << 32) | (threadIdx.x * 40503u + 77u);         \
    uint32_t b = threadIdx.x * 97u + seed;                                                                    \
    _Pragma("unroll") for (int i = 0; i < UNROLL; i++) acc[i] = i;                                            \
    for (int it = 0; it < ITERS; it++) {                                                                      \
        _Pragma("unroll") for (int i = 0; i < UNROLL; i++)                                                    \
            asm volatile(ASM : "+v"(acc[i]) : "v"(a), "v"(b));                                                \
    }                                                                                                         \
    uint64_t s = 0;                                                                                           \
    _Pragma("unroll") for (int i = 0; i < UNROLL; i++) s ^= acc[i];                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                           \
}

#define KERNEL32(NAME, ASM)                                                                                   \
__global__ void NAME(uint64_t* out, uint32_t seed) {                                                          \
    uint32_t acc[UNROLL];                                                                                     \
    uint32_t a = threadIdx.x * 2654435761u + seed;                                                            \
    uint32_t b = threadIdx.x * 97u + seed;                                                                    \
    _Pragma("unroll") for (int i = 0; i < UNROLL; i++) acc[i] = i;                                            \
    for (int it = 0; it < ITERS; it++) {                                                                      \
        _Pragma("unroll") for (int i = 0; i < UNROLL; i++)                                                    \
            asm volatile(ASM : "+v"(acc[i]) : "v"(a), "v"(b));                                                \
    }                                                                                                         \
    uint32_t s = 0;                                                                                           \
    _Pragma("unroll") for (int i = 0; i < UNROLL; i++) s ^= acc[i];                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                           \
}

KERNEL64(k_qsad,   "v_qsad_pk_u16_u8 %0, %1, %2, %0")
KERNEL64(k_mqsad,  "v_mqsad_pk_u16_u8 %0, %1, %2, %0")
KERNEL32(k_sad_u8, "v_sad_u8 %0, %1, %2, %0")
KERNEL32(k_msad_u8,"v_msad_u8 %0, %1, %2, %0")
KERNEL32(k_sad_u16,"v_sad_u16 %0, %1, %2, %0")
KERNEL32(k_sad_u32,"v_sad_u32 %0, %1, %2, %0")
KERNEL32(k_pk_add, "v_pk_add_u16 %0, %0, %1")
KERNEL32(k_pk_sub, "v_pk_sub_u16 %0, %0, %1")
KERNEL32(k_pk_min, "v_pk_min_u16 %0, %0, %1")
KERNEL32(k_pk_mad, "v_pk_mad_u16 %0, %0, %1, %2")
KERNEL32(k_min3,   "v_min3_u32 %0, %0, %1, %2")
KERNEL32(k_min_u32,"v_min_u32 %0, %0, %1")
KERNEL32(k_add_u32,"v_add_u32 %0, %0, %1")
KERNEL32(k_lshl_or,"v_lshl_or_b32 %0, %0, 8, %1")
KERNEL32(k_alignbyte,"v_alignbyte_b32 %0, %0, %1, %2")
KERNEL32(k_perm,   "v_perm_b32 %0, %0, %1, %2")
KERNEL32(k_min_u16,"v_min_u16 %0, %0, %1")
KERNEL32(k_dpp_min,"v_min_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_mad_u32_u24,"v_mad_u32_u24 %0, %1, %2, %0")
KERNEL32(k_sub_sdwa,"v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2")

KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %1")
KERNEL32(k_sub_u32, "v_sub_u32 %0, %0, %1")
KERNEL32(k_and_b32, "v_and_b32 %0, %0, %1")
KERNEL32(k_or_b32, "v_or_b32 %0, %0, %1")
KERNEL32(k_lshlrev, "v_lshlrev_b32 %0, 3, %0")
KERNEL32(k_bfe, "v_bfe_u32 %0, %0, 8, 8")
KERNEL32(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
KERNEL32(k_add_dpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
KERNEL32(k_add_wshr, "v_add_u32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
KERNEL32(k_sub_wshl, "v_sub_u32_dpp %0, %1, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
KERNEL32(k_pl16swap, "v_permlane16_swap_b32 %0, %1")
KERNEL32(k_pl32swap, "v_permlane32_swap_b32 %0, %1")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_cmp_addc, "v_cmp_le_u32 vcc, %1, %0\n v_addc_co_u32 %0, vcc, 0, %0, vcc")
KERNEL32(k_min_sdwa, "v_min_u32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cndmask_s, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL32(k_cmp_cnd, "v_cmp_le_u32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cmp_only, "v_cmp_le_u32 vcc, %1, %0")
KERNEL32(k_cmp_s, "v_cmp_le_u32_e64 s[20:21], %1, %0")
KERNEL32(k_max_i16, "v_max_i16 %0, %0, %1")
KERNEL32(k_med3, "v_med3_u32 %0, %0, %1, %2")
KERNEL32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_max_u32, "v_max_u32 %0, %0, %1")
KERNEL32(k_min_i32, "v_min_i32 %0, %0, %1")
KERNEL32(k_xor, "v_xor_b32 %0, %0, %1")

struct Entry { const char* name; void (*fn)(uint64_t*, uint32_t); };

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    const int threads = 256, blocks = cus * 8;   // 8 waves per SIMD
    uint64_t* out; CK(hipMalloc(&out, (size_t)threads * blocks * 8));
    Entry es[] = {{"v_qsad_pk_u16_u8", k_qsad}, {"v_mqsad_pk_u16_u8", k_mqsad}, {"v_sad_u8", k_sad_u8}, {"v_msad_u8", k_msad_u8},
                  {"v_sad_u16", k_sad_u16}, {"v_sad_u32", k_sad_u32}, {"v_pk_add_u16", k_pk_add}, {"v_pk_sub_u16", k_pk_sub},
                  {"v_pk_min_u16", k_pk_min}, {"v_pk_mad_u16", k_pk_mad}, {"v_min3_u32", k_min3}, {"v_min_u32", k_min_u32},
                  {"v_add_u32", k_add_u32}, {"v_lshl_or_b32", k_lshl_or}, {"v_alignbyte_b32", k_alignbyte}, {"v_perm_b32", k_perm},
                  {"v_min_u16", k_min_u16}, {"v_min_u32_dpp", k_dpp_min}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_sub_u32_sdwa", k_sub_sdwa},
                  {"v_lshl_add_u64", k_lshl_add_u64}, {"v_sub_u32", k_sub_u32}, {"v_and_b32", k_and_b32}, {"v_or_b32", k_or_b32}, {"v_lshlrev_b32", k_lshlrev},
                  {"v_bfe_u32", k_bfe}, {"v_mov_b32_dpp", k_mov_dpp}, {"v_add_u32_dpp", k_add_dpp}, {"v_add_u32_dpp wave_shr:1", k_add_wshr}, {"v_sub_u32_dpp wave_shl:1", k_sub_wshl}, {"v_permlane16_swap", k_pl16swap}, {"v_permlane32_swap", k_pl32swap}, {"v_add3_u32", k_add3},
                  {"v_cmp_le+v_addc_co (2)", k_cmp_addc}, {"v_min_u32_sdwa", k_min_sdwa}, {"v_cndmask_b32", k_cndmask}, {"v_cndmask_b32_e64 sgpr", k_cndmask_s}, {"v_cmp_le+v_cndmask (2)", k_cmp_cnd}, {"v_cmp_le vcc", k_cmp_only},
                  {"v_cmp_le_e64 sgpr", k_cmp_s}, {"v_max_i16", k_max_i16}, {"v_med3_u32", k_med3}, {"v_mul_lo_u32", k_mul_lo},
                  {"v_mul_u32_u24", k_mul_u24}, {"v_max_u32", k_max_u32}, {"v_min_i32", k_min_i32}, {"v_xor_b32", k_xor}};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& e : es) {
        e.fn<<<blocks, threads>>>(out, 1); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; r++) e.fn<<<blocks, threads>>>(out, r);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        double winstr = (double)blocks * (threads / 64) * ITERS * UNROLL;          // wave-instructions
        double per_simd_cyc = ms * 1e-3 * 2.4e9 / (winstr / (cus * 4.0));         // cycles per wave-instr per SIMD @2.4GHz
        printf("%-22s %8.3f ms  %6.2f cyc/wave-instr/SIMD (assuming 2.4 GHz)  %7.2f T lane-instr/s\n", e.name, ms, per_simd_cyc,
               winstr * 64 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
