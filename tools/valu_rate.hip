// VALU issue-rate probe for the integer / packed-16 / byte instructions the ORB front-end is made of:
// cycles per wave-instruction on one SIMD, with 1, 2 and 4 wavefronts resident on that SIMD (a workgroup of 256 / 512 / 1024
// threads on one compute unit), eight independent destination registers per instruction stream.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o tools/_bin/valu_rate && tools/_bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// OP3: "op dst, src0, src1, dst"-shaped (three sources), OP2: two sources, OP1: one source
#define KERNEL(name, STR)                                                                                                     \
    __global__ void name(long long* out, unsigned* sink)                                                                       \
    {                                                                                                                          \
        unsigned a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 0x01020304u + threadIdx.x, c = 0x00030001u; \
        __syncthreads();                                                                                                       \
        const long long t0 = clock64();                                                                                        \
        REP64(asm volatile(STR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc", "s10", "s11");) \
        const long long t1 = clock64();                                                                                        \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                                                          \
        sink[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                             \
    }
#define S3(op) op " %0, %8, %9, %0\n" op " %1, %8, %9, %1\n" op " %2, %8, %9, %2\n" op " %3, %8, %9, %3\n" op " %4, %8, %9, %4\n" op " %5, %8, %9, %5\n" op " %6, %8, %9, %6\n" op " %7, %8, %9, %7"
#define S2(op) op " %0, %8, %0\n" op " %1, %8, %1\n" op " %2, %8, %2\n" op " %3, %8, %3\n" op " %4, %8, %4\n" op " %5, %8, %5\n" op " %6, %8, %6\n" op " %7, %8, %7"
#define S2I(op) op " %0, %8, %9\n" op " %1, %8, %9\n" op " %2, %8, %9\n" op " %3, %8, %9\n" op " %4, %8, %9\n" op " %5, %8, %9\n" op " %6, %8, %9\n" op " %7, %8, %9"
#define S1(op) op " %0, %8\n" op " %1, %8\n" op " %2, %8\n" op " %3, %8\n" op " %4, %8\n" op " %5, %8\n" op " %6, %8\n" op " %7, %8"
#define SCMP(op) op " vcc, %8, %0\n" op " vcc, %8, %1\n" op " vcc, %8, %2\n" op " vcc, %8, %3\n" op " vcc, %8, %4\n" op " vcc, %8, %5\n" op " vcc, %8, %6\n" op " vcc, %8, %7"
#define SCND(op) op " %0, %8, %0, vcc\n" op " %1, %8, %1, vcc\n" op " %2, %8, %2, vcc\n" op " %3, %8, %3, vcc\n" op " %4, %8, %4, vcc\n" op " %5, %8, %5, vcc\n" op " %6, %8, %6, vcc\n" op " %7, %8, %7, vcc"

KERNEL(k_add_f32, S2("v_add_f32"))
KERNEL(k_min_f32, S2("v_min_f32"))
KERNEL(k_fma_f32, S3("v_fma_f32"))
KERNEL(k_max3_f32, S3("v_max3_f32"))
KERNEL(k_add_u32, S2("v_add_u32"))
KERNEL(k_sub_u32, S2("v_sub_u32"))
KERNEL(k_and_b32, S2("v_and_b32"))
KERNEL(k_xor_b32, S2("v_xor_b32"))
KERNEL(k_lshrrev, S2("v_lshrrev_b32"))
KERNEL(k_min_i32, S2("v_min_i32"))
KERNEL(k_min_u32, S2("v_min_u32"))
KERNEL(k_min3_i32, S3("v_min3_i32"))
KERNEL(k_med3_i32, S3("v_med3_i32"))
KERNEL(k_min_u16, S2("v_min_u16"))
KERNEL(k_sub_u16, S2("v_sub_u16"))
KERNEL(k_pk_min_i16, S2("v_pk_min_i16"))
KERNEL(k_pk_max_u16, S2("v_pk_max_u16"))
KERNEL(k_pk_sub_i16, S2("v_pk_sub_i16"))
KERNEL(k_pk_add_u16, S2("v_pk_add_u16"))
KERNEL(k_pk_mad_u16, S3("v_pk_mad_u16"))
KERNEL(k_pk_mul_lo_u16, S2("v_pk_mul_lo_u16"))
KERNEL(k_pk_min_f16, S2("v_pk_min_f16"))
KERNEL(k_pk_add_f16, S2("v_pk_add_f16"))
KERNEL(k_perm, S3("v_perm_b32"))
KERNEL(k_alignbyte, S3("v_alignbyte_b32"))
KERNEL(k_alignbit, S3("v_alignbit_b32"))
KERNEL(k_bfi, S3("v_bfi_b32"))
KERNEL(k_bfe, S3("v_bfe_u32"))
KERNEL(k_and_or, S3("v_and_or_b32"))
KERNEL(k_lshl_or, S3("v_lshl_or_b32"))
KERNEL(k_lshl_add, S3("v_lshl_add_u32"))
KERNEL(k_add3, S3("v_add3_u32"))
KERNEL(k_dot4_u8, S3("v_dot4_u32_u8"))
KERNEL(k_dot2_u16, S3("v_dot2_u32_u16"))
KERNEL(k_mad_u24, S3("v_mad_u32_u24"))
KERNEL(k_mul_u24, S2("v_mul_u32_u24"))
KERNEL(k_mul_lo_u32, S2("v_mul_lo_u32"))
KERNEL(k_sad_u8, S3("v_sad_u8"))
KERNEL(k_msad_u8, S3("v_msad_u8"))
KERNEL(k_sad_u16, S3("v_sad_u16"))
KERNEL(k_bcnt, S2("v_bcnt_u32_b32"))
KERNEL(k_mbcnt_lo, S2("v_mbcnt_lo_u32_b32"))
KERNEL(k_mov, S1("v_mov_b32"))
KERNEL(k_cvt_ubyte0, S1("v_cvt_f32_ubyte0"))
KERNEL(k_or_b32, S2("v_or_b32"))
KERNEL(k_not_b32, S1("v_not_b32"))
KERNEL(k_max_u16, S2("v_max_u16"))
KERNEL(k_min_i16, S2("v_min_i16"))
KERNEL(k_max_i16, S2("v_max_i16"))
KERNEL(k_add_u16, S2("v_add_u16"))
KERNEL(k_lshlrev, S2("v_lshlrev_b32"))
KERNEL(k_ashrrev, S2("v_ashrrev_i32"))
KERNEL(k_max_i32, S2("v_max_i32"))
KERNEL(k_max_f32, S2("v_max_f32"))
KERNEL(k_mul_f32, S2("v_mul_f32"))
KERNEL(k_sub_f32, S2("v_sub_f32"))
KERNEL(k_subrev_u32, S2("v_subrev_u32"))
KERNEL(k_mad_i24, S3("v_mad_i32_i24"))
KERNEL(k_bfe_i32, S3("v_bfe_i32"))
KERNEL(k_xad, S3("v_xad_u32"))
KERNEL(k_or3, S3("v_or3_b32"))
KERNEL(k_bitop3, "v_bitop3_b32 %0, %8, %9, %0 bitop3:0x96\n v_bitop3_b32 %1, %8, %9, %1 bitop3:0x96\n v_bitop3_b32 %2, %8, %9, %2 bitop3:0x96\n v_bitop3_b32 %3, %8, %9, %3 bitop3:0x96\n v_bitop3_b32 %4, %8, %9, %4 bitop3:0x96\n v_bitop3_b32 %5, %8, %9, %5 bitop3:0x96\n v_bitop3_b32 %6, %8, %9, %6 bitop3:0x96\n v_bitop3_b32 %7, %8, %9, %7 bitop3:0x96")
KERNEL(k_add_u32_e64, S2("v_add_u32_e64"))
KERNEL(k_and_b32_e64, S2("v_and_b32_e64"))
KERNEL(k_min_u16_e64, S2("v_min_u16_e64"))
KERNEL(k_add_sdwa, "v_add_u32_sdwa %0, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %3, %8, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %4, %8, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %5, %8, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %6, %8, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %7, %8, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_cmp_gt_i32, SCMP("v_cmp_gt_i32"))
KERNEL(k_cmp_gt_u16, SCMP("v_cmp_gt_u16"))
KERNEL(k_cndmask_s, "v_cndmask_b32_e64 %0, %8, %0, s[10:11]\n v_cndmask_b32_e64 %1, %8, %1, s[10:11]\n v_cndmask_b32_e64 %2, %8, %2, s[10:11]\n v_cndmask_b32_e64 %3, %8, %3, s[10:11]\n v_cndmask_b32_e64 %4, %8, %4, s[10:11]\n v_cndmask_b32_e64 %5, %8, %5, s[10:11]\n v_cndmask_b32_e64 %6, %8, %6, s[10:11]\n v_cndmask_b32_e64 %7, %8, %7, s[10:11]")
KERNEL(k_cndmask2, "v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc")
KERNEL(k_cndmask, SCND("v_cndmask_b32"))

KERNEL(k_pair_vcc, "v_cmp_gt_i32 vcc, %8, %0\n v_cndmask_b32 %0, %8, %9, vcc\n v_cmp_gt_i32 vcc, %8, %1\n v_cndmask_b32 %1, %8, %9, vcc\n v_cmp_gt_i32 vcc, %8, %2\n v_cndmask_b32 %2, %8, %9, vcc\n v_cmp_gt_i32 vcc, %8, %3\n v_cndmask_b32 %3, %8, %9, vcc\n v_cmp_gt_i32 vcc, %8, %4\n v_cndmask_b32 %4, %8, %9, vcc\n v_cmp_gt_i32 vcc, %8, %5\n v_cndmask_b32 %5, %8, %9, vcc\n v_cmp_gt_i32 vcc, %8, %6\n v_cndmask_b32 %6, %8, %9, vcc\n v_cmp_gt_i32 vcc, %8, %7\n v_cndmask_b32 %7, %8, %9, vcc")
KERNEL(k_pair_sgpr, "v_cmp_gt_i32_e64 s[10:11], %8, %0\n v_cndmask_b32_e64 %0, %8, %9, s[10:11]\n v_cmp_gt_i32_e64 s[10:11], %8, %1\n v_cndmask_b32_e64 %1, %8, %9, s[10:11]\n v_cmp_gt_i32_e64 s[10:11], %8, %2\n v_cndmask_b32_e64 %2, %8, %9, s[10:11]\n v_cmp_gt_i32_e64 s[10:11], %8, %3\n v_cndmask_b32_e64 %3, %8, %9, s[10:11]\n v_cmp_gt_i32_e64 s[10:11], %8, %4\n v_cndmask_b32_e64 %4, %8, %9, s[10:11]\n v_cmp_gt_i32_e64 s[10:11], %8, %5\n v_cndmask_b32_e64 %5, %8, %9, s[10:11]\n v_cmp_gt_i32_e64 s[10:11], %8, %6\n v_cndmask_b32_e64 %6, %8, %9, s[10:11]\n v_cmp_gt_i32_e64 s[10:11], %8, %7\n v_cndmask_b32_e64 %7, %8, %9, s[10:11]")
KERNEL(k_cmp_e64, "v_cmp_gt_i32_e64 s[10:11], %8, %0\n v_cmp_gt_i32_e64 s[10:11], %8, %1\n v_cmp_gt_i32_e64 s[10:11], %8, %2\n v_cmp_gt_i32_e64 s[10:11], %8, %3\n v_cmp_gt_i32_e64 s[10:11], %8, %4\n v_cmp_gt_i32_e64 s[10:11], %8, %5\n v_cmp_gt_i32_e64 s[10:11], %8, %6\n v_cmp_gt_i32_e64 s[10:11], %8, %7")
KERNEL(k_add_co, "v_add_co_u32 %0, vcc, %8, %0\n v_add_co_u32 %1, vcc, %8, %1\n v_add_co_u32 %2, vcc, %8, %2\n v_add_co_u32 %3, vcc, %8, %3\n v_add_co_u32 %4, vcc, %8, %4\n v_add_co_u32 %5, vcc, %8, %5\n v_add_co_u32 %6, vcc, %8, %6\n v_add_co_u32 %7, vcc, %8, %7")
KERNEL(k_addc_co, "v_addc_co_u32 %0, vcc, %8, %0, vcc\n v_addc_co_u32 %1, vcc, %8, %1, vcc\n v_addc_co_u32 %2, vcc, %8, %2, vcc\n v_addc_co_u32 %3, vcc, %8, %3, vcc\n v_addc_co_u32 %4, vcc, %8, %4, vcc\n v_addc_co_u32 %5, vcc, %8, %5, vcc\n v_addc_co_u32 %6, vcc, %8, %6, vcc\n v_addc_co_u32 %7, vcc, %8, %7, vcc")
KERNEL(k_and_lit, "v_and_b32 %0, 0x00ff00ff, %0\n v_and_b32 %1, 0x00ff00ff, %1\n v_and_b32 %2, 0x00ff00ff, %2\n v_and_b32 %3, 0x00ff00ff, %3\n v_and_b32 %4, 0x00ff00ff, %4\n v_and_b32 %5, 0x00ff00ff, %5\n v_and_b32 %6, 0x00ff00ff, %6\n v_and_b32 %7, 0x00ff00ff, %7")
KERNEL(k_sub_lit, "v_sub_u32 %0, 0x12345678, %0\n v_sub_u32 %1, 0x12345678, %1\n v_sub_u32 %2, 0x12345678, %2\n v_sub_u32 %3, 0x12345678, %3\n v_sub_u32 %4, 0x12345678, %4\n v_sub_u32 %5, 0x12345678, %5\n v_sub_u32 %6, 0x12345678, %6\n v_sub_u32 %7, 0x12345678, %7")
KERNEL(k_lshl_b16, "v_lshlrev_b16 %0, 3, %0\n v_lshlrev_b16 %1, 3, %1\n v_lshlrev_b16 %2, 3, %2\n v_lshlrev_b16 %3, 3, %3\n v_lshlrev_b16 %4, 3, %4\n v_lshlrev_b16 %5, 3, %5\n v_lshlrev_b16 %6, 3, %6\n v_lshlrev_b16 %7, 3, %7")
KERNEL(k_lshr_b16, "v_lshrrev_b16 %0, 3, %0\n v_lshrrev_b16 %1, 3, %1\n v_lshrrev_b16 %2, 3, %2\n v_lshrrev_b16 %3, 3, %3\n v_lshrrev_b16 %4, 3, %4\n v_lshrrev_b16 %5, 3, %5\n v_lshrrev_b16 %6, 3, %6\n v_lshrrev_b16 %7, 3, %7")
KERNEL(k_pk_lshr_b16, "v_pk_lshrrev_b16 %0, 3, %0\n v_pk_lshrrev_b16 %1, 3, %1\n v_pk_lshrrev_b16 %2, 3, %2\n v_pk_lshrrev_b16 %3, 3, %3\n v_pk_lshrrev_b16 %4, 3, %4\n v_pk_lshrrev_b16 %5, 3, %5\n v_pk_lshrrev_b16 %6, 3, %6\n v_pk_lshrrev_b16 %7, 3, %7")
KERNEL(k_mul_lo_u16, "v_mul_lo_u16 %0, %8, %0\n v_mul_lo_u16 %1, %8, %1\n v_mul_lo_u16 %2, %8, %2\n v_mul_lo_u16 %3, %8, %3\n v_mul_lo_u16 %4, %8, %4\n v_mul_lo_u16 %5, %8, %5\n v_mul_lo_u16 %6, %8, %6\n v_mul_lo_u16 %7, %8, %7")
KERNEL(k_mad_u16, "v_mad_u16 %0, %8, %9, %0\n v_mad_u16 %1, %8, %9, %1\n v_mad_u16 %2, %8, %9, %2\n v_mad_u16 %3, %8, %9, %3\n v_mad_u16 %4, %8, %9, %4\n v_mad_u16 %5, %8, %9, %5\n v_mad_u16 %6, %8, %9, %6\n v_mad_u16 %7, %8, %9, %7")
KERNEL(k_cvt_pk_u8, "v_cvt_pk_u8_f32 %0, %8, %9, %0\n v_cvt_pk_u8_f32 %1, %8, %9, %1\n v_cvt_pk_u8_f32 %2, %8, %9, %2\n v_cvt_pk_u8_f32 %3, %8, %9, %3\n v_cvt_pk_u8_f32 %4, %8, %9, %4\n v_cvt_pk_u8_f32 %5, %8, %9, %5\n v_cvt_pk_u8_f32 %6, %8, %9, %6\n v_cvt_pk_u8_f32 %7, %8, %9, %7")
KERNEL(k_readlane, "v_readlane_b32 s10, %0, 3\n v_readlane_b32 s10, %1, 3\n v_readlane_b32 s10, %2, 3\n v_readlane_b32 s10, %3, 3\n v_readlane_b32 s10, %4, 3\n v_readlane_b32 s10, %5, 3\n v_readlane_b32 s10, %6, 3\n v_readlane_b32 s10, %7, 3")

typedef int v4i __attribute__((ext_vector_type(4)));
typedef long v2l __attribute__((ext_vector_type(2)));
#define MFMA_KERNEL(name, CALL, NPER)                                                                  \
    __global__ void name(long long* out, unsigned* sink)                                               \
    {                                                                                                  \
        v4i c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0;                                            \
        long a = threadIdx.x * 0x0101010101010101l; v4i b4 = { (int)threadIdx.x, 1, 2, 3 };            \
        __syncthreads();                                                                               \
        const long long t0 = clock64();                                                                \
        for (int i = 0; i < 64; ++i) { CALL }                                                          \
        const long long t1 = clock64();                                                                \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = (t1 - t0) * 512 / (64 * NPER);            \
        sink[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];                                             \
    }
MFMA_KERNEL(k_mfma_i8_32, c0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, a, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, a, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, a, c3, 0, 0, 0);, 4)
MFMA_KERNEL(k_mfma_i8_64, c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b4, b4, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b4, b4, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b4, b4, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(b4, b4, c3, 0, 0, 0);, 4)
MFMA_KERNEL(k_mfma_i8_32_dep, c0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, a, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, a, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, a, c0, 0, 0, 0); c0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, a, c0, 0, 0, 0);, 4)

#define KERNEL64(name, STR)                                                                                                   \
    __global__ void name(long long* out, unsigned* sink)                                                                       \
    {                                                                                                                          \
        unsigned long long a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, b = 0x0102030405060708ull + threadIdx.x; unsigned c0 = 0, c1 = 0, c2 = 0, c3 = 0; \
        __syncthreads();                                                                                                       \
        const long long t0 = clock64();                                                                                        \
        REP64(asm volatile(STR : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b) : "vcc");) \
        const long long t1 = clock64();                                                                                        \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = (t1 - t0);                                                        \
        sink[threadIdx.x] = c0 + c1 + c2 + c3;                                                                                 \
    }
// four compares (vcc) per asm block, each followed by an add-with-carry into its own counter: 8 instructions per block
KERNEL64(k_cmp_u64, "v_cmp_gt_u64 vcc, %4, %8\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_cmp_gt_u64 vcc, %5, %8\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n v_cmp_gt_u64 vcc, %6, %8\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_cmp_gt_u64 vcc, %7, %8\n v_addc_co_u32 %3, vcc, 0, %3, vcc")
KERNEL64(k_cmp_f64, "v_cmp_gt_f64 vcc, %4, %8\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_cmp_gt_f64 vcc, %5, %8\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n v_cmp_gt_f64 vcc, %6, %8\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_cmp_gt_f64 vcc, %7, %8\n v_addc_co_u32 %3, vcc, 0, %3, vcc")

#define KERNEL64D(name, STR)                                                                                                  \
    __global__ void name(long long* out, unsigned* sink)                                                                       \
    {                                                                                                                          \
        unsigned long long a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7; unsigned b = 12345u + threadIdx.x, c = 77u; \
        __syncthreads();                                                                                                       \
        const long long t0 = clock64();                                                                                        \
        REP64(asm volatile(STR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");) \
        const long long t1 = clock64();                                                                                        \
        if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = (t1 - t0);                                                        \
        sink[threadIdx.x] = (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                                                 \
    }
KERNEL64D(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7")
KERNEL64D(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 3, %1\n v_lshl_add_u64 %1, %1, 3, %2\n v_lshl_add_u64 %2, %2, 3, %3\n v_lshl_add_u64 %3, %3, 3, %4\n v_lshl_add_u64 %4, %4, 3, %5\n v_lshl_add_u64 %5, %5, 3, %6\n v_lshl_add_u64 %6, %6, 3, %7\n v_lshl_add_u64 %7, %7, 3, %0")
KERNEL64D(k_lshlrev_b64, "v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 3, %1\n v_lshlrev_b64 %2, 3, %2\n v_lshlrev_b64 %3, 3, %3\n v_lshlrev_b64 %4, 3, %4\n v_lshlrev_b64 %5, 3, %5\n v_lshlrev_b64 %6, 3, %6\n v_lshlrev_b64 %7, 3, %7")
KERNEL64D(k_add_f64, "v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_add_f64 %2, %2, %3\n v_add_f64 %3, %3, %4\n v_add_f64 %4, %4, %5\n v_add_f64 %5, %5, %6\n v_add_f64 %6, %6, %7\n v_add_f64 %7, %7, %0")
KERNEL64D(k_mul_f64, "v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_mul_f64 %3, %3, %4\n v_mul_f64 %4, %4, %5\n v_mul_f64 %5, %5, %6\n v_mul_f64 %6, %6, %7\n v_mul_f64 %7, %7, %0")
KERNEL64D(k_fma_f64, "v_fma_f64 %0, %1, %2, %0\n v_fma_f64 %1, %2, %3, %1\n v_fma_f64 %2, %3, %4, %2\n v_fma_f64 %3, %4, %5, %3\n v_fma_f64 %4, %5, %6, %4\n v_fma_f64 %5, %6, %7, %5\n v_fma_f64 %6, %7, %0, %6\n v_fma_f64 %7, %0, %1, %7")
KERNEL64D(k_mov_b64, "v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %4\n v_mov_b64 %4, %5\n v_mov_b64 %5, %6\n v_mov_b64 %6, %7\n v_mov_b64 %7, %0")

struct Entry { const char* name; void (*fn)(long long*, unsigned*); };

int main()
{
    long long* d; unsigned* s; hipMalloc(&d, 16 * 8); hipMalloc(&s, 4 * 1024);
    const Entry es[] = {
        { "v_add_f32", k_add_f32 }, { "v_min_f32", k_min_f32 }, { "v_fma_f32", k_fma_f32 }, { "v_max3_f32", k_max3_f32 },
        { "v_add_u32", k_add_u32 }, { "v_sub_u32", k_sub_u32 }, { "v_and_b32", k_and_b32 }, { "v_xor_b32", k_xor_b32 }, { "v_lshrrev_b32", k_lshrrev },
        { "v_min_i32", k_min_i32 }, { "v_min_u32", k_min_u32 }, { "v_min3_i32", k_min3_i32 }, { "v_med3_i32", k_med3_i32 },
        { "v_min_u16", k_min_u16 }, { "v_sub_u16", k_sub_u16 },
        { "v_pk_min_i16", k_pk_min_i16 }, { "v_pk_max_u16", k_pk_max_u16 }, { "v_pk_sub_i16", k_pk_sub_i16 }, { "v_pk_add_u16", k_pk_add_u16 },
        { "v_pk_mad_u16", k_pk_mad_u16 }, { "v_pk_mul_lo_u16", k_pk_mul_lo_u16 }, { "v_pk_min_f16", k_pk_min_f16 }, { "v_pk_add_f16", k_pk_add_f16 },
        { "v_perm_b32", k_perm }, { "v_alignbyte_b32", k_alignbyte }, { "v_alignbit_b32", k_alignbit }, { "v_bfi_b32", k_bfi }, { "v_bfe_u32", k_bfe },
        { "v_and_or_b32", k_and_or }, { "v_lshl_or_b32", k_lshl_or }, { "v_lshl_add_u32", k_lshl_add }, { "v_add3_u32", k_add3 },
        { "v_dot4_u32_u8", k_dot4_u8 }, { "v_dot2_u32_u16", k_dot2_u16 }, { "v_mad_u32_u24", k_mad_u24 }, { "v_mul_u32_u24", k_mul_u24 }, { "v_mul_lo_u32", k_mul_lo_u32 },
        { "v_sad_u8", k_sad_u8 }, { "v_msad_u8", k_msad_u8 }, { "v_sad_u16", k_sad_u16 }, { "v_bcnt_u32_b32", k_bcnt }, { "v_mbcnt_lo_u32_b32", k_mbcnt_lo },
        { "v_mov_b32", k_mov }, { "v_cvt_f32_ubyte0", k_cvt_ubyte0 }, { "v_or_b32", k_or_b32 }, { "v_not_b32", k_not_b32 }, { "v_max_u16", k_max_u16 }, { "v_min_i16", k_min_i16 }, { "v_max_i16", k_max_i16 }, { "v_add_u16", k_add_u16 },
        { "v_lshlrev_b32", k_lshlrev }, { "v_ashrrev_i32", k_ashrrev }, { "v_max_i32", k_max_i32 }, { "v_max_f32", k_max_f32 }, { "v_mul_f32", k_mul_f32 }, { "v_sub_f32", k_sub_f32 }, { "v_subrev_u32", k_subrev_u32 },
        { "v_mad_i32_i24", k_mad_i24 }, { "v_bfe_i32", k_bfe_i32 }, { "v_xad_u32", k_xad }, { "v_or3_b32", k_or3 }, { "v_bitop3_b32", k_bitop3 },
        { "v_add_u32_e64 (VOP3)", k_add_u32_e64 }, { "v_and_b32_e64 (VOP3)", k_and_b32_e64 }, { "v_min_u16_e64 (VOP3)", k_min_u16_e64 }, { "v_add_u32_sdwa", k_add_sdwa }, { "v_mov_b32_dpp row_shr", k_mov_dpp },
        { "v_cmp_gt_i32 (vcc)", k_cmp_gt_i32 }, { "v_cmp_gt_u16 (vcc)", k_cmp_gt_u16 }, { "v_cndmask_b32_e64 sgpr", k_cndmask_s }, { "v_cndmask_b32 dst!=src", k_cndmask2 }, { "v_cndmask_b32", k_cndmask },
        { "PAIR v_cmp vcc + v_cndmask vcc (per pair)", k_pair_vcc }, { "PAIR v_cmp_e64 sgpr + v_cndmask_e64 (per pair)", k_pair_sgpr }, { "v_cmp_gt_i32_e64 sgpr", k_cmp_e64 },
        { "v_add_co_u32 vcc", k_add_co }, { "v_addc_co_u32 vcc", k_addc_co }, { "v_and_b32 literal", k_and_lit }, { "v_sub_u32 literal", k_sub_lit }, 
        { "v_lshlrev_b16", k_lshl_b16 }, { "v_lshrrev_b16", k_lshr_b16 }, { "v_pk_lshrrev_b16", k_pk_lshr_b16 }, { "v_mul_lo_u16", k_mul_lo_u16 }, { "v_mad_u16", k_mad_u16 }, { "v_cvt_pk_u8_f32", k_cvt_pk_u8 }, { "v_readlane_b32", k_readlane },
        { "v_mad_u64_u32", k_mad_u64_u32 }, { "v_lshl_add_u64", k_lshl_add_u64 }, { "v_lshlrev_b64", k_lshlrev_b64 }, { "v_add_f64", k_add_f64 }, { "v_mul_f64", k_mul_f64 }, { "v_fma_f64", k_fma_f64 }, { "v_mov_b64", k_mov_b64 },
        { "PAIR v_cmp_gt_u64 + v_addc (per pair)", k_cmp_u64 }, { "PAIR v_cmp_gt_f64 + v_addc (per pair)", k_cmp_f64 },
        { "v_mfma_i32_16x16x32_i8", k_mfma_i8_32 }, { "v_mfma_i32_16x16x64_i8", k_mfma_i8_64 }, { "v_mfma_i32_16x16x32_i8 dependent", k_mfma_i8_32_dep },
    };
    printf("%-24s %8s %8s %8s   (cycles per wave-instruction per SIMD at 1 / 2 / 4 wavefronts per SIMD)\n", "instruction", "1", "2", "4");
    for (const Entry& e : es) {
        double r[3];
        for (int wi = 0; wi < 3; ++wi) {
            const int waves_per_simd = 1 << wi, threads = 256 * waves_per_simd;
            for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(e.fn, dim3(1), dim3(threads), 0, 0, d, s); hipDeviceSynchronize(); }
            long long o[16]; hipMemcpy(o, d, 8 * (threads / 64), hipMemcpyDeviceToHost);
            long long mx = 0; for (int i = 0; i < threads / 64; ++i) mx = o[i] > mx ? o[i] : mx;
            r[wi] = (double)mx / 512.0 / waves_per_simd;
        }
        printf("%-24s %8.2f %8.2f %8.2f\n", e.name, r[0], r[1], r[2]);
    }
    return 0;
}
