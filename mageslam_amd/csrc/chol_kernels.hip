// chol_kernels.hip -- dense float64 Cholesky factorisation + solve of the reduced camera system on gfx950.
//
// Replaces g2o::LinearSolverDense::solve (Eigen::LDLT<MatrixXd>), which
// BundlerLib selects at Dependencies/BundlerLib/Source/BundlerLib.cpp:188-190.  The reduced camera
// matrix S is symmetric positive definite whenever the reference's LDLT reports isPositive(), so an
// unpivoted Cholesky S = L L^T gives the same solution up to rounding; a non-positive pivot is
// reported through *ok = 0 (the reference's "solve failed" branch).
//
// Layout: S is n_pad x n_pad, column-major, lower triangle referenced, n_pad a multiple of TILE = 128.
// Right-looking tile algorithm, one panel of TILE columns per step k:
//     k_potrf_diag   : L_kk = chol(S_kk), plus the inverses of its eight 16x16 diagonal blocks
//     k_trsm_panel   : L_ik = S_ik L_kk^-T for i > k, and the rhs row  y_k = L_kk^-1 y_k
//     k_syrk_update  : S_ij -= L_ik L_jk^T  (i >= j > k), and  y_i -= L_ik y_k
// so the forward substitution rides along with the factorisation (the rhs is one more matrix row).
// The backward substitution L^T x = y is k_bsolve_persist, one persistent launch.
//
// All O(n^3) work is on the f64 matrix cores (v_mfma_f64_16x16x4_f64, 64 cycles / instruction / SIMD).
// Two properties of that instruction shape the kernels:
//   * operand A is lane -> A[i = lane & 15][k = lane >> 4], operand B is lane -> B[k = lane >> 4][j = lane & 15],
//     and the result register r of lane holds D[row = (lane >> 4) + 4 r][col = lane & 15];
//   * hence a 16x16 result D, taken register by register, IS the B operand of the next product
//     (register r = rows 4r..4r+3 = the r-th K-chunk).  Every triangular solve below is written on the
//     transposed unknown (Y = X^T) so that results chain from MFMA to MFMA without leaving registers.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "chol_kernels.h"
#include <atomic>
#include <string>

namespace mage {
namespace {

constexpr int TILE = CHOL_TILE;      // 128
constexpr int NB = 16;               // inner block (one MFMA tile)
constexpr int NBLK = TILE / NB;      // 8
constexpr int LDC = TILE + 16;       // LDS column pitch (doubles): consecutive k columns land 32 banks apart
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr unsigned long long X_SENTINEL = 0x7FF4DEADBEEF0001ull;   // "not published yet" in x (k_bsolve_persist): a signalling-NaN pattern no computation produces

__device__ __forceinline__ double readlane_d(double v, int lane)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// sqrt(d) and 1/sqrt(d) by v_rsq_f64 + two Goldschmidt steps (full double precision for d > 0)
__device__ __forceinline__ void sqrt_rsqrt(double d, double& s, double& rs)
{
    double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    s = g; rs = h + h;
}

// Global (column-major, pitch ld) -> LDS (column-major, pitch LDC) copy of one 128x128 tile by 256 threads.
// Loads are issued 16 at a time per thread (16-byte each) so that the L2/HBM latency is paid 4 times, not 64.
template <int PITCH>
__device__ __forceinline__ void load_tile(double* __restrict__ dst, const double* __restrict__ src, int ld, int tid)
{
    constexpr int BATCH = 16;
#pragma unroll
    for (int b0 = 0; b0 < (TILE * TILE / 2) / 256; b0 += BATCH) {
        double2 v[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = ((b0 + u) * 256 + tid) * 2;
            v[u] = *reinterpret_cast<const double2*>(src + (size_t)(e / TILE) * ld + (e % TILE));
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = ((b0 + u) * 256 + tid) * 2;
            *reinterpret_cast<double2*>(dst + (e / TILE) * PITCH + (e % TILE)) = v[u];
        }
    }
}

// Two LDS layouts of a diagonal tile, both column-major inside a 16x16 block:
//   LayLDC     the whole 128 x 128 square with column pitch LDC (147 KB): the small dense solve, which also substitutes out of it;
//   LayPacked  the 36 blocks of the lower triangle behind each other (block (rb, cb), rb >= cb, at (rb (rb + 1) / 2 + cb) * 256):
//              72 KB, so that TWO workgroups of the trailing update fit a compute unit beside the one that factors the next
//              diagonal tile (LDS is sized per launch, not per workgroup).  Register r of lane l of an MFMA operand / accumulator
//              is element 64 r + l of its block: every wave-wide LDS access is 512 contiguous bytes.
struct LayLDC {
    static constexpr int PITCH = LDC;
    static __device__ __forceinline__ int blk(int rb, int cb) { return cb * NB * LDC + rb * NB; }
};
struct LayPacked {
    static constexpr int PITCH = NB;
    static __device__ __forceinline__ int blk(int rb, int cb) { return (rb * (rb + 1) / 2 + cb) * (NB * NB); }
};
constexpr int PACKED_TILE_DOUBLES = (NBLK * (NBLK + 1) / 2) * NB * NB;      // 9216

// ---------------------------------------------------------------------------------------------
// 16x16 diagonal block at A(p0, p0): Cholesky in registers by one wavefront, together with the inverse of the factor.
// Lane l (mod 16; the four 16-lane rows of the wavefront run identical copies) owns ROW l of the block (a[c] = A[l][c])
// and COLUMN l of L^-1 (x[c] = Linv[c][l]).  Both recurrences are
//     a[c] -= a[j] * L[c][j],   x[c] -= x[j] * L[c][j]      (c > j),        a[j], x[j] *= 1 / L[j][j]
// and the coefficient L[c][j] is lane c's a[j]: it enters the FMA as a DPP operand (row_newbcast:c -- the only DPP
// control gfx90a+ allows on 64-bit operations, and exactly the one needed), so an update is ONE v_fmac_f64_dpp with no
// trip through the SGPR file (the v_readlane form stalled on the VALU->SGPR->VALU round trip: 490 cycles per pivot).
//
// The pivot recurrence is the critical path of the whole factorisation (128 sequential pivots per tile; a dependent f64
// operation costs ~25 cycles on a lone wavefront), so it is cut to the bone -- 8 dependent operations per pivot:
//     d_p = e0_p - q4_p * h_{p-1}^2          e0_p = A[p][p] after the updates of columns <= p-2            (off the chain)
//                                            q4_p = (2 A[p][p-1])^2, same state, i.e. L[p][p-1] = sqrt(q4_p) h_{p-1}
//     h_p = 1 / (2 sqrt(d_p))                v_rsq_f64 + two Goldschmidt steps
// Column scaling (a[j] = 2 a[j] * h_j; the diagonal entry becomes d * rsqrt(d), no select), the DPP updates and the
// broadcasts that prepare e0 / q4 of the pivot after next are independent work, issued BETWEEN the chain's operations
// (a lone in-order wavefront hides latency no other way).  sched_barriers pin that interleaving.
// Inline asm is invisible to the hazard recogniser: "VALU writes VGPR -> DPP reads it" needs 2 wait states, supplied by
// explicit s_nop where a DPP source was written just before.
// A non-positive pivot is reported (ballot) and leaves NaNs behind; the caller discards the factorisation.
// Writes L over the block, the inverse row-major to Li and Linv_out.
// ---------------------------------------------------------------------------------------------
template <int C>
__device__ __forceinline__ void dpp_fnma(double& acc, double bsrc, double mul)        // acc -= bsrc[lane C of the row] * mul
{
    asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(mul), "n"(C));
}
template <int C>
__device__ __forceinline__ void dpp_fnma_nop(double& acc, double bsrc, double mul)    // same, bsrc written just before
{
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(mul), "n"(C));
}
template <int C>
__device__ __forceinline__ double dpp_bcast(double v)                                  // v[lane C of the row]
{
    double r;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(C));
    return r;
}
template <int C>
__device__ __forceinline__ double dpp_bcast_nop(double v)                              // same, v written just before
{
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(C));
    return r;
}

template <int J, int C>
__device__ __forceinline__ void fb_update(double (&a)[NB], double (&x)[NB])
{
    if constexpr (C < NB) {
        dpp_fnma<C>(a[C], a[J], a[J]);
        dpp_fnma<C>(x[C], a[J], x[J]);
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int J, int C>
__device__ __forceinline__ void fb_update_rest(double (&a)[NB], double (&x)[NB])
{
    if constexpr (C < NB) {
        fb_update<J, C>(a, x);
        fb_update_rest<J, C + 1>(a, x);
    }
}

// Column J.  In: h = h_J, (q4, e0) of pivot J + 1, a2 / x2 = twice the final unscaled column J.  Out: the same for J + 1.
template <int J>
__device__ __forceinline__ void fb_column(double (&a)[NB], double (&x)[NB], double& h, double& q4, double& e0, double& a2, double& x2, double& dmin)
{
    double a2n = 0, x2n = 0, q4n = 0, e0n = 0, hn = h;
    // chain step 1                                   | side: scale column J
    const double hh = h * h;
    a[J] = a2 * h;                                     // L[:, J]  (lane J: d * rsqrt(d) = sqrt(d))
    x[J] = x2 * h;                                     // Linv[J][:] (exactly 0 for lanes l > J: x2 is)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (J + 1 < NB) {
        // chain step 2: next pivot                    | side: the two updates the pivot after next depends on
        double dn = __builtin_fma(-q4, hh, e0);
        dpp_fnma_nop<J + 1>(a[J + 1], a[J], a[J]);
        dpp_fnma<J + 1>(x[J + 1], a[J], x[J]);
        __builtin_amdgcn_sched_barrier(0);
        double y = __builtin_amdgcn_rsq(dn);
        if constexpr (J + 2 < NB) {
            dpp_fnma<J + 2>(a[J + 2], a[J], a[J]);
            dpp_fnma<J + 2>(x[J + 2], a[J], x[J]);
        }
        dmin = fmin(dmin, dn);                         // a non-positive pivot is caught at the end (NaNs only follow one)
        __builtin_amdgcn_sched_barrier(0);
        double g = dn * y; hn = 0.5 * y;
        a2n = a[J + 1] + a[J + 1];                     // column J + 1 is final (unscaled) now
        x2n = x[J + 1] + x[J + 1];
        __builtin_amdgcn_sched_barrier(0);
        double r = __builtin_fma(-hn, g, 0.5);
        double t2 = 0;
        if constexpr (J + 2 < NB) {
            t2 = dpp_bcast_nop<J + 2>(a2n);            // 2 A[J+2][J+1]
            e0n = dpp_bcast<J + 2>(a[J + 2]);          // A[J+2][J+2] after the updates of columns <= J
        }
        __builtin_amdgcn_sched_barrier(0);
        g = __builtin_fma(g, r, g); hn = __builtin_fma(hn, r, hn);
        q4n = t2 * t2;
        fb_update<J, J + 3>(a, x);
        r = __builtin_fma(-hn, g, 0.5);
        __builtin_amdgcn_sched_barrier(0);
        fb_update<J, J + 4>(a, x);
        g = __builtin_fma(g, r, g); hn = __builtin_fma(hn, r, hn);
        __builtin_amdgcn_sched_barrier(0);
        fb_update_rest<J, J + 5>(a, x);
    }
    h = hn; q4 = q4n; e0 = e0n; a2 = a2n; x2 = x2n;
}

// B: the block (element (r, c) at B[c * PITCH + r])
template <int PITCH>
__device__ __forceinline__ bool factor_block16(double* __restrict__ B, int lane, double* __restrict__ Li, double* __restrict__ Linv_out)
{
    const int l = lane & 15;
    double a[NB], x[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) { a[c] = B[c * PITCH + l]; x[c] = (l == c) ? 1.0 : 0.0; }
    const double d0 = dpp_bcast_nop<0>(a[0]);
    double dmin = d0;
    double sq, rs;
    sqrt_rsqrt(d0, sq, rs);
    double h = 0.5 * rs, a2 = a[0] + a[0], x2 = x[0] + x[0];
    const double t2 = dpp_bcast_nop<1>(a2);
    double e0 = dpp_bcast<1>(a[1]);
    double q4 = t2 * t2;
    (void)sq;
    fb_column<0>(a, x, h, q4, e0, a2, x2, dmin);   fb_column<1>(a, x, h, q4, e0, a2, x2, dmin);   fb_column<2>(a, x, h, q4, e0, a2, x2, dmin);
    fb_column<3>(a, x, h, q4, e0, a2, x2, dmin);   fb_column<4>(a, x, h, q4, e0, a2, x2, dmin);   fb_column<5>(a, x, h, q4, e0, a2, x2, dmin);
    fb_column<6>(a, x, h, q4, e0, a2, x2, dmin);   fb_column<7>(a, x, h, q4, e0, a2, x2, dmin);   fb_column<8>(a, x, h, q4, e0, a2, x2, dmin);
    fb_column<9>(a, x, h, q4, e0, a2, x2, dmin);   fb_column<10>(a, x, h, q4, e0, a2, x2, dmin);  fb_column<11>(a, x, h, q4, e0, a2, x2, dmin);
    fb_column<12>(a, x, h, q4, e0, a2, x2, dmin);  fb_column<13>(a, x, h, q4, e0, a2, x2, dmin);  fb_column<14>(a, x, h, q4, e0, a2, x2, dmin);
    fb_column<15>(a, x, h, q4, e0, a2, x2, dmin);
    // The strict upper triangle of the block is left holding partial sums: nothing reads it (the panel solves use the
    // stored inverse for diagonal blocks, the updates only touch blocks below them, S's upper triangle is never referenced).
    if (lane < NB) {
#pragma unroll
        for (int c = 0; c < NB; ++c) B[c * PITCH + l] = a[c];
#pragma unroll
        for (int i = 0; i < NB; ++i) Li[i * NB + l] = x[i];
    }
    (void)Linv_out;    // copied from Li by the caller's other wavefronts, off the critical path
    return __builtin_amdgcn_ballot_w64(!(dmin > 0.0)) != 0;
}

// ---------------------------------------------------------------------------------------------
// The same 16 x 16 pivot block in its LEAN form (round 4, late).  tools/f64_latency.hip: on a lone wavefront a dependent v_fma_f64
// returns after 8 cycles, v_rsq_f64 after 20, a DPP operand after 16 -- and ANY f64 operation, DPP or not, takes 4.8 cycles of issue.
// The pivot recurrence of the form above is ~76 cycles of latency per pivot against ~40 instructions x 4.8 = 190 cycles of issue: the
// block is bound by instruction COUNT, not by the chain the form above was shortened for (two-pivots-ahead e0 / q4, doubled columns,
// broadcasts: ~620 instructions per block).  Here a pivot is the plain recurrence -- broadcast the diagonal entry, rsqrt + two
// Goldschmidt steps, scale the column, update the columns behind it -- 13 + 2 (15 - J) instructions, ~450 per block; the previous
// pivot's updates are dealt into the latency slots of this pivot's chain (one after every dependent operation, four behind the rsqrt).
// Same layout, same outputs as factor_block16.
// ---------------------------------------------------------------------------------------------
template <int P, int C>
__device__ __forceinline__ void fl_upd_a(double (&a)[NB])
{
    if constexpr (P >= 0 && C < NB) dpp_fnma<C>(a[C], a[P], a[P]);
}
template <int P, int C>
__device__ __forceinline__ void fl_upd_x(double (&a)[NB], double (&x)[NB])
{
    if constexpr (P >= 0 && C < NB) dpp_fnma<C>(x[C], a[P], x[P]);
}
// filler number F (0, 1, 2 ...) of pivot J's latency slots: the updates of pivot P = J - 1 from column J + 1 on, `x` of J + 1 first
// (its `a` update was issued at the end of pivot P: the diagonal entry of pivot J depends on it)
template <int J, int F>
__device__ __forceinline__ void fl_fill(double (&a)[NB], double (&x)[NB])
{
    constexpr int P = J - 1;
    if constexpr (F == 0) fl_upd_x<P, J + 0 + 0>(a, x);          // column J itself: x[J] (a[J] went first)
    else {
        constexpr int C = J + (F + 1) / 2;
        if constexpr ((F & 1) == 1) fl_upd_a<P, C>(a); else fl_upd_x<P, C>(a, x);
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int J, int F0, int F1>
__device__ __forceinline__ void fl_fill_range(double (&a)[NB], double (&x)[NB])
{
    if constexpr (F0 < F1) { fl_fill<J, F0>(a, x); fl_fill_range<J, F0 + 1, F1>(a, x); }
}
template <int J>
__device__ __forceinline__ void fl_column(double (&a)[NB], double (&x)[NB], double& dmin)
{
    constexpr int NFILL = J >= 1 ? 2 * (NB - J) - 1 : 0;         // fillers of pivot J - 1: x[J], then (a, x) of columns J + 1 .. 15
    // the diagonal entry (lane J's a[J], final: pivot J - 1 updated it last thing) to every lane
    const double d = dpp_bcast_nop<J>(a[J]);
    __builtin_amdgcn_sched_barrier(0);
    fl_fill_range<J, 0, (NFILL < 2 ? NFILL : 2)>(a, x);
    double y = __builtin_amdgcn_rsq(d);
    __builtin_amdgcn_sched_barrier(0);
    fl_fill_range<J, 2, (NFILL < 6 ? NFILL : 6)>(a, x);
    (void)dmin;                                                  // (a non-positive pivot leaves NaNs behind: caught once, at the end)
    // 1 / sqrt(d) from the ~2^-22 estimate in ONE third-order step -- e = 1 - d y^2, rs = y (1 + e / 2 + 3 e^2 / 8), error ~ 5/16 e^3 < 2^-60 --
    // five dependent operations behind the estimate where two Goldschmidt steps were six, and three instructions fewer per pivot
    const double t = d * y;
    __builtin_amdgcn_sched_barrier(0);
    fl_fill_range<J, 6, (NFILL < 7 ? NFILL : 7)>(a, x);
    const double e = __builtin_fma(-t, y, 1.0);
    __builtin_amdgcn_sched_barrier(0);
    fl_fill_range<J, 7, (NFILL < 8 ? NFILL : 8)>(a, x);
    const double p = __builtin_fma(e, 0.375, 0.5);
    __builtin_amdgcn_sched_barrier(0);
    fl_fill_range<J, 8, (NFILL < 9 ? NFILL : 9)>(a, x);
    const double q = e * p;
    __builtin_amdgcn_sched_barrier(0);
    fl_fill_range<J, 9, (NFILL < 11 ? NFILL : 11)>(a, x);
    const double rs = __builtin_fma(y, q, y);                    // 1 / sqrt(d)
    __builtin_amdgcn_sched_barrier(0);
    fl_fill_range<J, 11, (NFILL < 12 ? NFILL : 12)>(a, x);
    a[J] = a[J] * rs;                                            // L[:, J]  (lane J: d / sqrt(d))
    x[J] = x[J] * rs;                                            // Linv[J][:]
    __builtin_amdgcn_sched_barrier(0);
    // the next pivot's diagonal entry first, then what is left of pivot J - 1's updates
    if constexpr (J + 1 < NB) dpp_fnma_nop<J + 1>(a[J + 1], a[J], a[J]);
    __builtin_amdgcn_sched_barrier(0);
    fl_fill_range<J, 12, (NFILL > 12 ? NFILL : 12)>(a, x);
}
template <int J>
__device__ __forceinline__ void fl_columns(double (&a)[NB], double (&x)[NB], double& dmin)
{
    if constexpr (J < NB) { fl_column<J>(a, x, dmin); fl_columns<J + 1>(a, x, dmin); }
}
template <int PITCH>
__device__ __forceinline__ bool factor_block16_lean(double* __restrict__ B, int lane, double* __restrict__ Li, double* __restrict__ Linv_out)
{
    const int l = lane & 15;
    double a[NB], x[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) { a[c] = B[c * PITCH + l]; x[c] = (l == c) ? 1.0 : 0.0; }
    double dmin = 1.0;
    fl_columns<0>(a, x, dmin);
    // (pivot 15's updates: none; pivot 14's leftovers were dealt inside pivot 15)
    // A pivot d <= 0 makes rsq(d) NaN or infinite, its column NaN (0 x inf for d = 0), and every later column of the rows below it NaN:
    // the last diagonal entry (lane 15's a[15]) is NaN exactly when some pivot was not positive -- sixteen v_min_f64 less on the
    // wavefront whose instruction count is the tile's critical path.
    dmin = (l == NB - 1 && !(a[NB - 1] == a[NB - 1])) ? -1.0 : 1.0;
    if (lane < NB) {
#pragma unroll
        for (int c = 0; c < NB; ++c) B[c * PITCH + l] = a[c];
#pragma unroll
        for (int i = 0; i < NB; ++i) Li[i * NB + l] = x[i];
    }
    (void)Linv_out;
    return __builtin_amdgcn_ballot_w64(!(dmin > 0.0)) != 0;
}

// which form the tile factorisation uses: the lean one (tools/potrf_probe.hip: 3 600 against 3 950 cycles per block, the tile 43.9 k
// against 45.3 k; tools/_bin/chol_test 6016: 2.480-2.491 against 2.495 ms).  Late in round 4 its 1 / sqrt(d) became ONE third-order step
// (fl_column): 3 660 -> 3 516 cycles per block, the tile 43.9 k -> 42.5 k, 2.485 -> 2.468-2.480 ms; its bits now differ from the classic block's
// in the last place (residual 7.89e-16 against 7.90e-16, |Linv L - I| 2.2e-16 against 3.3e-16).
// -DCHOL_FACTOR_BLOCK=factor_block16 builds the classic one.
#ifndef CHOL_FACTOR_BLOCK
#define CHOL_FACTOR_BLOCK factor_block16_lean
#endif

// element of register r of an MFMA operand / accumulator inside a block
template <class LAY>
__device__ __forceinline__ int frag(int r, int lane) { return (4 * r + (lane >> 4)) * LAY::PITCH + (lane & 15); }

// one 16x16 block of the in-LDS trailing update: block (bi, bj) -= X(bi, bp) X(bj, bp)^T.
// D[m][n] = C[row n][col m] of the block: the accumulator's lane&15 direction is the LDS-contiguous one.
template <class LAY>
__device__ __forceinline__ void lds_update_tile(double* __restrict__ A, int bi, int bj, int bp, int lane)
{
    double* C = A + LAY::blk(bi, bj);
    const double* Xa = A + LAY::blk(bj, bp);
    const double* Xb = A + LAY::blk(bi, bp);
    double4_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = C[frag<LAY>(r, lane)];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double aop = -Xa[frag<LAY>(r, lane)];
        const double bop = Xb[frag<LAY>(r, lane)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) C[frag<LAY>(r, lane)] = acc[r];
}

// Left-looking form of the same update: block (bi, bj) -= sum over the nk column blocks kb = 0 .. nk - 1 of
// X(bi, kb) X(bj, kb)^T, accumulator loaded and stored once, operand reads of block kb + 1 in flight
// while block kb's MFMAs run.
template <class LAY>
__device__ __forceinline__ void lds_update_tile_left(double* __restrict__ A, int bi, int bj, int nk, int lane)
{
    double* C = A + LAY::blk(bi, bj);
    double4_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = C[frag<LAY>(r, lane)];
    double aop[2][4], bop[2][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        aop[0][r] = -A[LAY::blk(bj, 0) + frag<LAY>(r, lane)];
        bop[0][r] = A[LAY::blk(bi, 0) + frag<LAY>(r, lane)];
    }
    for (int kb = 0; kb < nk; kb += 2) {
        if (kb + 1 < nk) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                aop[1][r] = -A[LAY::blk(bj, kb + 1) + frag<LAY>(r, lane)];
                bop[1][r] = A[LAY::blk(bi, kb + 1) + frag<LAY>(r, lane)];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[0][r], bop[0][r], acc, 0, 0, 0);
        if (kb + 1 < nk) {
            if (kb + 2 < nk) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    aop[0][r] = -A[LAY::blk(bj, kb + 2) + frag<LAY>(r, lane)];
                    bop[0][r] = A[LAY::blk(bi, kb + 2) + frag<LAY>(r, lane)];
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[1][r], bop[1][r], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) C[frag<LAY>(r, lane)] = acc[r];
}

// ---------------------------------------------------------------------------------------------
// diagonal tile, LDS-resident, blocked by 16.  Per block s: the rows below are solved on the matrix
// cores with the block inverse (Y = Linv A^T); then wavefront 0 updates only the NEXT diagonal block and
// factors it while wavefronts 1-3 apply the rest of the trailing update (look-ahead inside the tile).
// ---------------------------------------------------------------------------------------------
// Factor the LDS-resident tile A (column-major, pitch LDC) in place; Li = 2 x 256 doubles of LDS scratch.
// PARTIAL: only the leading nblk 16-column blocks are factored (the rest of the tile is the identity padding of a small system).
// PUBLISH (the merged, pipelined panel solve of k_syrk_update): the factoring workgroup makes its progress visible to the strips of
// the same launch block column by block column instead of tile by tile -- with no flag and no wait.  At the top of iteration s
// (diagonal block s factored, block column s - 1 final) wavefront 3 writes the inverse of block s to its place in the workspace and
// wavefronts 1-3 deal out the blocks (i, s - 1), i >= s, to a scratch copy of the tile's sub-diagonal blocks (LPUB_TILE_DOUBLES per
// tile, block (c, j) at (c (c - 1) / 2 + j) * 256, column-major), all with agent-scope stores that go THROUGH to memory.  Both areas
// were filled with X_SENTINEL when the factorisation started (k_trsm_panel of column 0), and a strip simply polls the operand values
// it is about to use until none of them is the sentinel -- what the backward solve does with x.  Waiting for the stores to drain
// before raising a flag made the publishing wavefront late at the iteration's barrier (+2 ... +6 us per tile on the critical path).
constexpr int LPUB_BLOCKS = NBLK * (NBLK - 1) / 2;            // 28 sub-diagonal blocks
constexpr int LPUB_TILE_DOUBLES = LPUB_BLOCKS * NB * NB;      // 7168
struct TilePublish {
    double* Lpub;                 // PUBLISH == 2: this tile's scratch blocks (nullptr: not publishing)
                                  // PUBLISH == 3: the tile's full inverse is built in the PACKED_TILE_DOUBLES of LDS right behind Li's two blocks, blocks ROW-major
    double* inv_global = nullptr; // PUBLISH == 3: where the inverse goes, block (c, j), c >= j, at (c (c + 1) / 2 + j) * 256, blocks COLUMN-major
    int* progress = nullptr;      // PUBLISH == 4: one word: base + c once the blocks of columns <= c (and the block inverses <= c + 1) are in memory
    int base = 0;                 //               (8 x tile index: the words only ever grow inside a factorisation)
};

// PUBLISH: 0 plain stores of the block inverses (a kernel boundary or a release fence follows); 1 the block inverses written THROUGH
// (agent-scope stores: the write-through hand-off of k_syrk_update<1, true>, no release fence follows); 2 also the sub-diagonal blocks
// (the pipelined strips described above); 3 (round 4) as 1, and the tile's FULL INVERSE L^-1 (lower triangular, 36 blocks) is built
// alongside and written through, so that the panel solve below the tile is a product, X = A L^-T, with no dependent chain (trsm_strip_gemm):
//     Linv[i][j] = -Linv[i][i] P_i[j],   P_i[j] = sum_{k = j .. i - 1} L[i][k] Linv[k][j]        (i > j; from L Linv = I)
// Column j of the inverse depends only on itself, so wavefront 1 + j % 3 owns it and nobody synchronises: during the factorisation of
// diagonal block s + 1 (wavefront 0, ~2 us) the owner finishes row s of its columns -- four matrix-core operations per block with the
// block inverse that has just become known -- and prepares P_{s+1}[j], which needs nothing newer than block row s + 1 of column s.
// What is left behind the LAST diagonal block is four operations per block of the last row.  Blocks live row-major in LDS (the B
// operand of the next product) and go to memory column-major (the A operand the strips read, 512 contiguous bytes per fragment):
// the transposed block is the same product with the operands exchanged, P^T (-Linv_ii)^T, from the registers already loaded.
// The barriers inside the tile factorisation order LDS traffic only.  __syncthreads() also waits for the wavefront's outstanding
// GLOBAL stores (the compiler puts s_waitcnt vmcnt(0) in front of the barrier), and in the publishing forms those are write-through
// stores that take ~2 us to be acknowledged: with the full inverse streaming out block by block every one of the tile's 24 barriers
// waited for memory (the in-tile factorisation went from 22 to 37 us).  LDS_ONLY: wait for this wavefront's LDS operations, then the
// hardware barrier; global stores stay in flight (the caller drains them once, before it raises the flag).
template <bool LDS_ONLY>
__device__ __forceinline__ void tile_barrier()
{
    if (LDS_ONLY) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
}

// A write-through (agent-scope) store to GLOBAL memory, typed as such: through a generic pointer it is a flat store, and the compiler
// must then assume it may hit LDS -- every later LDS access of the wavefront waits for vmcnt(0), i.e. for the ~2 us acknowledgement
// of a store that only ever goes to memory.
typedef __attribute__((address_space(1))) double global_double;
__device__ __forceinline__ void store_through(double* p, double v)
{
    __hip_atomic_store((global_double*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef CHOL_TILE_STAMPS        // tools/potrf_probe.hip: shader-clock stamps of every wavefront at the five points of an in-tile iteration
__device__ long long g_tile_stamps[4][NBLK][5];
#define TILE_STAMP(p) do { if (lane == 0) g_tile_stamps[wave][s][p] = clock64(); } while (0)
#else
#define TILE_STAMP(p) do { } while (0)
#endif

template <bool PARTIAL, class LAY, int PUBLISH = 0>
__device__ __forceinline__ bool potrf_tile_lds(double* __restrict__ A, double* __restrict__ Li, double* __restrict__ Linv_k, int tid, int nblk = NBLK,
                                            TilePublish pub = TilePublish{ nullptr })
{
    const int NBK = PARTIAL ? nblk : NBLK;
    const int lane = tid & 63, wave = tid >> 6;
    bool failed = false;
    if (wave == 0) failed = CHOL_FACTOR_BLOCK<LAY::PITCH>(A + LAY::blk(0, 0), lane, Li, Linv_k);
    tile_barrier<PUBLISH != 0>();
    for (int s = 0; s < NBK; ++s) {
        const double* Lc = Li + (s & 1) * NB * NB;
        TILE_STAMP(0);
        if (PUBLISH == 4 && wave >= 1 && s >= 2) {
            // PHASED strips (round 4): what this wavefront wrote through one iteration ago (its blocks of column s - 2, wavefront 3 also the
            // inverse of block s - 1) has been acknowledged by now, so the wait costs nothing -- unlike a flag raised right behind the stores,
            // which made the publisher late at the iteration's barrier.  Behind the iteration's first barrier (every wavefront has passed
            // this wait) wavefront 3 raises ONE progress word; the strips of the launch poll that word, not the operands.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (wave == 3) {                           // block inverse s -> global workspace (read by k_trsm_panel / k_bsolve_persist)
            if (PUBLISH) {
#pragma unroll
                for (int q = 0; q < 4; ++q) store_through(Linv_k + s * NB * NB + lane * 4 + q, Lc[lane * 4 + q]);
            } else {
                const double4_t v = *reinterpret_cast<const double4_t*>(Lc + lane * 4);
                *reinterpret_cast<double4_t*>(Linv_k + s * NB * NB + lane * 4) = v;
            }
            if (PUBLISH == 3) {                    // the diagonal block of the full inverse: row-major to LDS, column-major to memory
                const double4_t v = *reinterpret_cast<const double4_t*>(Lc + lane * 4);
                *reinterpret_cast<double4_t*>(Li + 2 * NB * NB + LAY::blk(s, s) + lane * 4) = v;      // (the inverse sits right behind the two block inverses: derived from Li, not handed in -- a second LDS pointer made the compiler look the LDS base up in memory every iteration)
                double* G = pub.inv_global + (size_t)(s * (s + 1) / 2 + s) * NB * NB;
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int e = lane * 4 + q; store_through(G + (e & 15) * NB + (e >> 4), v[q]); }
            }
        }
        if ((PUBLISH == 2 || PUBLISH == 4) && wave >= 1 && s > 0) {       // block column s - 1 is final: its blocks below the diagonal go out, dealt over wavefronts 1-3
            for (int i = s + wave - 1; i < NBK; i += 3) {
                const double* Bl = A + LAY::blk(i, s - 1);
                double* G = pub.Lpub + (size_t)(i * (i - 1) / 2 + (s - 1)) * NB * NB;
#pragma unroll
                for (int r = 0; r < 4; ++r) store_through(G + (4 * r + (lane >> 4)) * NB + (lane & 15), Bl[frag<LAY>(r, lane)]);
            }
        }
        if (s == NBK - 1) {
            if (PUBLISH == 4) {            // columns <= s - 2 are in memory (the waits above); no barrier follows in this iteration: one of its own
                tile_barrier<true>();
                if (wave == 3 && lane == 0) __hip_atomic_store(pub.progress, pub.base + s - 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            break;
        }
        // rows below block s:  Y = Linv * A^T per 16-row strip; Y[m][n] = X[row r0 + n][col p0 + m].
        // Wavefront 0 is the critical path: it solves only the strip it needs (the rows of the next diagonal block) and
        // updates that block before the barrier, while the other three share the remaining strips.
        const int nstrips = NBK - 1 - s;
        if (wave == 0) {
            // strip 0 and the next diagonal block in one go: the strip's result registers ARE both MFMA operands of
            // D -= Y^T Y (register r of a lane is element [4r + (lane >> 4)][lane & 15] of Y = operand chunk r of either side)
            double* Xs = A + LAY::blk(s + 1, s);          // strip 0: block row s + 1 of block column s
            double* Dn = A + LAY::blk(s + 1, s + 1);      // the next diagonal block
            double4_t acc = { 0, 0, 0, 0 }, dg;
            double aop[4], bop[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                aop[r] = Lc[(lane & 15) * NB + 4 * r + (lane >> 4)];
                bop[r] = Xs[frag<LAY>(r, lane)];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) dg[r] = Dn[frag<LAY>(r, lane)];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[r], bop[r], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Xs[frag<LAY>(r, lane)] = acc[r];
#pragma unroll
            for (int r = 0; r < 4; ++r) dg = __builtin_amdgcn_mfma_f64_16x16x4f64(-acc[r], acc[r], dg, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Dn[frag<LAY>(r, lane)] = dg[r];
            __builtin_amdgcn_s_waitcnt(0xc07f);
        } else {
            for (int t = wave; t < nstrips; t += 3) {
                double* Xs = A + LAY::blk(s + 1 + t, s);
                double4_t acc = { 0, 0, 0, 0 };
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double aop = Lc[(lane & 15) * NB + 4 * r + (lane >> 4)];
                    const double bop = Xs[frag<LAY>(r, lane)];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): all operand reads of this strip are done before it is overwritten
#pragma unroll
                for (int r = 0; r < 4; ++r) Xs[frag<LAY>(r, lane)] = acc[r];
            }
        }
        TILE_STAMP(1);
        tile_barrier<PUBLISH != 0>();
        TILE_STAMP(2);
        if (PUBLISH == 4 && s >= 2 && wave == 3 && lane == 0) __hip_atomic_store(pub.progress, pub.base + s - 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // Trailing update, scheduled so that it never outlasts the factorisation it runs beside (a right-looking update
        // front-loads 27 of the 77 tile updates into step 0; wavefront 0 then waited ~10k cycles per tile at this barrier):
        //   wavefront 0     factors the next diagonal block (updated just above);
        //   wavefronts 1-3  block column s+1 below the diagonal, LEFT-looking: tile (i, s+1) -= sum_{k <= s} Y_ik Y_{s+1,k}^T
        //                   (these are the strips of the next step), and the later diagonal blocks (i, i) -= Y_is Y_is^T.
        if (wave == 0) {
            failed |= CHOL_FACTOR_BLOCK<LAY::PITCH>(A + LAY::blk(s + 1, s + 1), lane, Li + ((s + 1) & 1) * NB * NB, Linv_k + (s + 1) * NB * NB);
        } else {
            const int nrow = NBK - 2 - s;                 // block rows s+2 .. 7
            // (with the inverse alongside, the update's tasks are dealt from wavefront 3 downwards: wavefront 1 owns the longest columns of the inverse)
            for (int t = PUBLISH == 3 ? 3 - wave : wave - 1; t < 2 * nrow; t += 3) {
                const int i = s + 2 + (t >> 1);
                if ((t & 1) == 0) lds_update_tile_left<LAY>(A, i, s + 1, s + 1, lane);
                else lds_update_tile<LAY>(A, i, i, s, lane);
            }
            if (PUBLISH == 3) {
                double* INV = Li + 2 * NB * NB;
                // columns dealt by cost (column j of row s + 1 is s + 1 - j products): 0 1 2 | 2 1 0 | 0 ... over wavefronts 1-3 -- a function
                // of j alone, so a column stays with its wavefront and nothing is handed over
                for (int j = 0; j <= s; ++j) {
                    const int z = j % 6;
                    if (1 + (z < 3 ? z : 5 - z) != wave) continue;
                    double4_t fin;                                        // Linv[s][j], accumulator layout = B operand of the next product
                    // operands of the first product of P_{s+1}[j] are fetched before anything else is waited for
                    double la[4], ib[4];
                    if (j < s) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { la[r] = A[LAY::blk(s + 1, j) + frag<LAY>(r, lane)]; ib[r] = INV[LAY::blk(j, j) + frag<LAY>(r, lane)]; }
                    }
                    if (j < s) {
                        double* slot = INV + LAY::blk(s, j);              // holds P_s[j]
                        double pr[4], li[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { pr[r] = slot[frag<LAY>(r, lane)]; li[r] = -Lc[(lane & 15) * NB + 4 * r + (lane >> 4)]; }
                        fin = (double4_t){ 0, 0, 0, 0 };
                        double4_t tr = { 0, 0, 0, 0 };
#pragma unroll
                        for (int r = 0; r < 4; ++r) fin = __builtin_amdgcn_mfma_f64_16x16x4f64(li[r], pr[r], fin, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) tr = __builtin_amdgcn_mfma_f64_16x16x4f64(pr[r], li[r], tr, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) slot[frag<LAY>(r, lane)] = fin[r];
                        double* G = pub.inv_global + (size_t)(s * (s + 1) / 2 + j) * NB * NB;
#pragma unroll
                        for (int r = 0; r < 4; ++r) store_through(G + frag<LAY>(r, lane), tr[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) fin[r] = Lc[frag<LAY>(r, lane)];
                    }
                    // P_{s+1}[j] = sum_{k = j .. s} L[s+1][k] Linv[k][j]  ->  the slot of block (s + 1, j); the operands of product k + 1 are in
                    // flight while product k runs
                    double4_t acc = { 0, 0, 0, 0 };
                    for (int kb = j; kb < s; ++kb) {
                        double la2[4], ib2[4];
                        if (kb + 1 < s) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { la2[r] = A[LAY::blk(s + 1, kb + 1) + frag<LAY>(r, lane)]; ib2[r] = INV[LAY::blk(kb + 1, j) + frag<LAY>(r, lane)]; }
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { la2[r] = A[LAY::blk(s + 1, s) + frag<LAY>(r, lane)]; ib2[r] = 0.0; }
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(la[r], ib[r], acc, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { la[r] = la2[r]; ib[r] = ib2[r]; }
                    }
                    if (j == s) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) la[r] = A[LAY::blk(s + 1, s) + frag<LAY>(r, lane)];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(la[r], fin[r], acc, 0, 0, 0);
                    double* nslot = INV + LAY::blk(s + 1, j);
#pragma unroll
                    for (int r = 0; r < 4; ++r) nslot[frag<LAY>(r, lane)] = acc[r];
                }
            }
        }
        TILE_STAMP(3);
        tile_barrier<PUBLISH != 0>();
        TILE_STAMP(4);
    }
    if (PUBLISH == 3) {
        // the last row of the inverse: Linv[7][j] = -Linv_77 P_7[j], straight to memory (transposed form only), two columns per wavefront
        const int sl = NBK - 1;
        const double* Lc = Li + (sl & 1) * NB * NB;
        for (int j = wave; j < sl; j += 4) {
            const double* slot = Li + 2 * NB * NB + LAY::blk(sl, j);
            double4_t tr = { 0, 0, 0, 0 };
#pragma unroll
            for (int r = 0; r < 4; ++r) tr = __builtin_amdgcn_mfma_f64_16x16x4f64(slot[frag<LAY>(r, lane)], -Lc[(lane & 15) * NB + 4 * r + (lane >> 4)], tr, 0, 0, 0);
            double* G = pub.inv_global + (size_t)(sl * (sl + 1) / 2 + j) * NB * NB;
#pragma unroll
            for (int r = 0; r < 4; ++r) store_through(G + frag<LAY>(r, lane), tr[r]);
        }
    }
    return failed;
}

// Global <-> LDS copies of a diagonal tile for LayPacked: the 36 lower blocks, 128-bit pieces along a block's columns, 18 per thread in
// two batches (the strict upper part of S is never read or written).
__device__ __forceinline__ void block_of_index(int t, int& rb, int& cb)          // t = rb (rb + 1) / 2 + cb, 0 <= t < 36
{
    rb = (t >= 1) + (t >= 3) + (t >= 6) + (t >= 10) + (t >= 15) + (t >= 21) + (t >= 28);
    cb = t - rb * (rb + 1) / 2;
}
__device__ __forceinline__ void load_tile_packed(double* __restrict__ dst, const double* __restrict__ src, int ld, int tid)
{
    constexpr int PIECES = PACKED_TILE_DOUBLES / 2, BATCH = 9;       // 4608 pieces = 256 threads x 18
#pragma unroll
    for (int b0 = 0; b0 < PIECES / 256; b0 += BATCH) {
        double2 v[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = (b0 + u) * 256 + tid, t = e >> 7, w = e & 127;      // block t, piece w: column w >> 3, rows 2 (w & 7) ..
            int rb, cb;
            block_of_index(t, rb, cb);
            v[u] = *reinterpret_cast<const double2*>(src + (size_t)(cb * NB + (w >> 3)) * ld + rb * NB + 2 * (w & 7));
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = (b0 + u) * 256 + tid;
            *reinterpret_cast<double2*>(dst + 2 * e) = v[u];
        }
    }
}
__device__ __forceinline__ void store_tile_packed(double* __restrict__ T, const double* __restrict__ A, int ld, int tid)
{
#pragma unroll 6
    for (int b0 = 0; b0 < PACKED_TILE_DOUBLES / 2 / 256; ++b0) {
        const int e = b0 * 256 + tid, t = e >> 7, w = e & 127;
        int rb, cb;
        block_of_index(t, rb, cb);
        *reinterpret_cast<double2*>(T + (size_t)(cb * NB + (w >> 3)) * ld + rb * NB + 2 * (w & 7)) = *reinterpret_cast<const double2*>(A + 2 * e);
    }
}

// The same two copies for a tile that is HANDED OVER inside a launch (k_syrk_update<1, true>): 128-bit buffer loads / stores with the
// sc1 bit.  An sc1 store goes through to memory and leaves no dirty line in this XCD's L2, so the producer needs no release fence (a
// release writes back whatever the XCD's L2 holds dirty -- here the update's freshly written tiles); an sc1 load is not served from
// this compute unit's L1, so the consumer of sc1-stored data needs no acquire fence (MI355X_MICROARCH.md, "Workgroup dispatch, XCD
// placement & inter-workgroup visibility").  T must be wave-uniform (it is: kernel arguments and blockIdx only).
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_tile_packed_wt(double* __restrict__ dst, const double* __restrict__ T, int ld, int tid)
{
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(T), 0, (int)((size_t)TILE * ld * sizeof(double)), 0x00020000);
    constexpr int PIECES = PACKED_TILE_DOUBLES / 2, BATCH = 9;
#pragma unroll
    for (int b0 = 0; b0 < PIECES / 256; b0 += BATCH) {
        u32x4_t v[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = (b0 + u) * 256 + tid, t = e >> 7, w = e & 127;
            int rb, cb;
            block_of_index(t, rb, cb);
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((size_t)(cb * NB + (w >> 3)) * ld + rb * NB + 2 * (w & 7)) * sizeof(double)), 0, 16);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = (b0 + u) * 256 + tid;
            *reinterpret_cast<u32x4_t*>(dst + 2 * e) = v[u];
        }
    }
}
__device__ __forceinline__ void store_tile_packed_wt(double* __restrict__ T, const double* __restrict__ A, int ld, int tid)
{
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(T, 0, (int)((size_t)TILE * ld * sizeof(double)), 0x00020000);
#pragma unroll 6
    for (int b0 = 0; b0 < PACKED_TILE_DOUBLES / 2 / 256; ++b0) {
        const int e = b0 * 256 + tid, t = e >> 7, w = e & 127;
        int rb, cb;
        block_of_index(t, rb, cb);
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4_t*>(A + 2 * e), rsrc,
                                               (int)(((size_t)(cb * NB + (w >> 3)) * ld + rb * NB + 2 * (w & 7)) * sizeof(double)), 0, 16);
    }
}

// ---------------------------------------------------------------------------------------------
// diagonal tile, LDS-resident, blocked by 16.  Per block s: the rows below are solved on the matrix
// cores with the block inverse (Y = Linv A^T); then wavefront 0 updates only the NEXT diagonal block and
// factors it while wavefronts 1-3 apply the rest of the trailing update (look-ahead inside the tile).
// Stand-alone form (first tile); later tiles are factored inside k_syrk_update (see there).
// ---------------------------------------------------------------------------------------------
// It also opens the solve: ok = 1, stall = 0, and x pre-filled with the sentinel the backward substitution polls for (they were
// three launches of their own in front of this one).
__global__ __launch_bounds__(256) void k_potrf_diag(double* __restrict__ S, int ld, int k, double* __restrict__ Linv_k,
                                                    double* __restrict__ ok, double* __restrict__ stall, unsigned long long* __restrict__ x_fill, int n_fill)
{
    extern __shared__ double sm[];
    double* A = sm;                       // LayPacked: the 36 lower blocks
    double* Li = sm + PACKED_TILE_DOUBLES;   // 2 x (16 x 16): inverse of the current / next diagonal block
    const int tid = threadIdx.x;
    if (tid == 0) { *ok = 1.0; *stall = 0.0; }
    for (int i = tid; i < n_fill; i += 256) x_fill[i] = X_SENTINEL;
    double* T = S + (size_t)(k * TILE) * ld + (size_t)k * TILE;
    load_tile_packed(A, T, ld, tid);
    __syncthreads();
    const bool failed = potrf_tile_lds<false, LayPacked>(A, Li, Linv_k, tid);
    store_tile_packed(T, A, ld, tid);
    if (tid == 0 && failed) *ok = 0.0;
}

// ---------------------------------------------------------------------------------------------
// panel solve  X L_kk^T = A  on the matrix cores.  One wavefront (= one workgroup) per 16-row strip,
// working on Y = X^T block by block:  Y_c = Linv_cc (A_c^T - sum_{j<c} L_cj Y_j).  L_kk and the block
// inverses are read straight from global memory (L2-resident, 128-byte segments in operand shape);
// Y stays in registers (result -> B operand identity).  No LDS, no barriers.  The last workgroup
// solves the rhs row y_k with the same code (a strip with one live row).
// ---------------------------------------------------------------------------------------------
// One strip of the panel solve of tile column k: strip < n_strips = a 16-row strip of the tiles below the diagonal, strip == n_strips =
// the rhs row y_k (a strip with one live row).
// inv_out != nullptr: the strip is rows 16 strip .. of the IDENTITY and the result, rows of L_kk^-T, goes to inv_out (element
// (row, col) at inv_out[col * inv_pitch + row]; the backward solve builds the tile's inverse this way, in LDS).
// (INVERSE is a template parameter so that the destination is an LDS pointer in one instantiation and a global one in the other: as
// a run-time choice it was a generic pointer, flat loads and stores)
template <bool INVERSE = false>
__device__ __forceinline__ void trsm_strip(double* __restrict__ S, double* __restrict__ y, int ld, int k, int strip, bool is_rhs,
                                           const double* __restrict__ Linv_k, int lane, double* inv_out = nullptr, int inv_pitch = 0)
{
    double* base;          // element (n = strip row, col) lives at base[col * cstride]; for the rhs strip only n == 0 exists
    size_t cstride;
    bool live;
    if (INVERSE) {
        base = inv_out + strip * NB + (lane & 15);
        cstride = (size_t)inv_pitch;
        live = true;
    } else if (!is_rhs) {
        base = S + (size_t)(k * TILE) * ld + (size_t)(k + 1) * TILE + strip * NB + (lane & 15);
        cstride = (size_t)ld;
        live = true;
    } else {
        base = y + (size_t)k * TILE;
        cstride = 1;
        live = (lane & 15) == 0;
    }
    double4_t Acc[NBLK];
#pragma unroll
    for (int c = 0; c < NBLK; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            Acc[c][r] = INVERSE ? ((strip * NB + (lane & 15)) == (c * NB + (lane >> 4) + 4 * r) ? 1.0 : 0.0)
                                : live ? base[(size_t)(c * NB + (lane >> 4) + 4 * r) * cstride] : 0.0;
    // operand (row = 16c + (lane&15), col = 16j + 4r + (lane>>4)) of L_kk
    const double* Lop = S + (size_t)(k * TILE + (lane >> 4)) * ld + (size_t)k * TILE + (lane & 15);
    const double* Lio = Linv_k + (lane & 15) * NB + (lane >> 4);
    // every operand is known up front: issue all loads, then run the MFMA chain
    double lop[NBLK][NBLK][4], lio[NBLK][4];
#pragma unroll
    for (int c = 0; c < NBLK; ++c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) lio[c][r] = Lio[c * NB * NB + 4 * r];
#pragma unroll
        for (int j = 0; j < c; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) lop[c][j][r] = -Lop[(size_t)(j * NB + 4 * r) * ld + c * NB];
    }
    double4_t Y[NBLK];
#pragma unroll
    for (int c = 0; c < NBLK; ++c) {
        double4_t acc = Acc[c];
#pragma unroll
        for (int j = 0; j < c; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lop[c][j][r], Y[j][r], acc, 0, 0, 0);
        double4_t yc = { 0, 0, 0, 0 };
#pragma unroll
        for (int r = 0; r < 4; ++r) yc = __builtin_amdgcn_mfma_f64_16x16x4f64(lio[c][r], acc[r], yc, 0, 0, 0);
        Y[c] = yc;
        if (live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) base[(size_t)(c * NB + (lane >> 4) + 4 * r) * cstride] = yc[r];
        }
    }
}

// The same strip for the MERGED panel solve, pipelined against the factorisation of L_kk that runs in the same launch (TilePublish):
// step c needs the inverse of diagonal block c and the blocks (c, j < c); it reads them from where the factoring workgroup writes them
// through to memory and polls each batch of four values until none is the sentinel.  Operands come through agent-scope loads (they
// bypass this XCD's L2).  What is left behind the last publication of a tile is one fetch, two products and a store.
__device__ __forceinline__ bool lane_has_sentinel(const double (&v)[4])
{
    bool pending = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) pending |= (unsigned long long)__double_as_longlong(v[r]) == X_SENTINEL;
    return pending;
}

__device__ __forceinline__ void trsm_strip_pipelined(double* __restrict__ S, double* __restrict__ y, int ld, int k, int strip, bool is_rhs,
                                                     const double* __restrict__ Linv_k, const double* __restrict__ Lpub, double* __restrict__ stall, int lane)
{
    double* base;
    size_t cstride;
    bool live;
    if (!is_rhs) {
        base = S + (size_t)(k * TILE) * ld + (size_t)(k + 1) * TILE + strip * NB + (lane & 15);
        cstride = (size_t)ld;
        live = true;
    } else {
        base = y + (size_t)k * TILE;
        cstride = 1;
        live = (lane & 15) == 0;
    }
    double4_t Acc[NBLK];
#pragma unroll
    for (int c = 0; c < NBLK; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) Acc[c][r] = live ? base[(size_t)(c * NB + (lane >> 4) + 4 * r) * cstride] : 0.0;
    // operand element (row = lane & 15, column = 4 r + (lane >> 4)) of a published block; of the row-major block inverse
    const double* Lop = Lpub + (lane >> 4) * NB + (lane & 15);
    const double* Lio = Linv_k + (lane & 15) * NB + (lane >> 4);
    // Every operand is fetched up front, as in the unpipelined strip: what has been published by now arrives with ONE memory
    // latency; what has not shows the sentinel and is polled for when its step comes.
    double lop[NBLK][NBLK][4], lio[NBLK][4];
#pragma unroll
    for (int c = 0; c < NBLK; ++c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) lio[c][r] = __hip_atomic_load(Lio + c * NB * NB + 4 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < c; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                lop[c][j][r] = __hip_atomic_load(Lop + (size_t)(c * (c - 1) / 2 + j) * NB * NB + (size_t)r * 4 * NB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double4_t Y[NBLK];
    bool ok = true;
#pragma unroll
    for (int c = 0; c < NBLK; ++c) {
        // the operands of THIS step, all together: re-read (after a pause -- two hundred strips polling back to back take a measurable
        // share of the fabric away from the factoring workgroup) until none of them shows the sentinel
        for (int spins = 0; ok; ++spins) {
            bool pending = lane_has_sentinel(lio[c]);
#pragma unroll
            for (int j = 0; j < c; ++j) pending |= lane_has_sentinel(lop[c][j]);
            if (!__any(pending)) break;
            if (spins >= (1 << 18)) { ok = false; break; }
            __builtin_amdgcn_s_sleep(8);
#pragma unroll
            for (int r = 0; r < 4; ++r) lio[c][r] = __hip_atomic_load(Lio + c * NB * NB + 4 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < c; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    lop[c][j][r] = __hip_atomic_load(Lop + (size_t)(c * (c - 1) / 2 + j) * NB * NB + (size_t)r * 4 * NB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        double4_t acc = Acc[c];
#pragma unroll
        for (int j = 0; j < c; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-lop[c][j][r], Y[j][r], acc, 0, 0, 0);
        double4_t yc = { 0, 0, 0, 0 };
#pragma unroll
        for (int r = 0; r < 4; ++r) yc = __builtin_amdgcn_mfma_f64_16x16x4f64(lio[c][r], acc[r], yc, 0, 0, 0);
        Y[c] = yc;
        if (live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) base[(size_t)(c * NB + (lane >> 4) + 4 * r) * cstride] = yc[r];
        }
    }
    if (!ok && lane == 0) *stall = 2.0;
}

// ---------------------------------------------------------------------------------------------
// trailing update.  One workgroup = one 128x128 tile of the lower triangle, 4 wavefronts as 2x2,
// each owning 64x64 = 4x4 MFMA tiles (128 accumulator registers).  Operands go straight from
// global memory to registers in MFMA fragment shape (16 consecutive rows x 4 panel columns per load:
// four 128-byte segments), prefetched one 32-column chunk ahead; no LDS, no barriers, every
// wavefront independent.  The MFMA "M" index runs over tile COLUMNS and "N" over tile ROWS so the
// accumulator's lane&15 direction is the memory-contiguous one; the accumulators are initialised
// with the C tile and the panel enters negated, so the epilogue is a plain store.
// ---------------------------------------------------------------------------------------------
// out[a][b][r] = C - sum_k L(col0.., k) L(row0.., k)^T for the (16 SUB) x (16 SUB) block whose first element is
// S(row0, col0); D layout: element (row0 + 16 b + (lane & 15), col0 + 16 a + (lane >> 4) + 4 r).
template <int SUBM, int SUBN, int KSTEPS, bool C_FIRST, int NBUF>
__device__ __forceinline__ void update_rect(const double* __restrict__ S, int ld, int k, int row0, int col0, int lane, double4_t (&out)[SUBM][SUBN]);

template <int SUB, int KSTEPS>
__device__ __forceinline__ void update_block(const double* __restrict__ S, int ld, int k, int row0, int col0, int lane, double4_t (&out)[SUB][SUB])
{
    const double* Pn = S + (size_t)(k * TILE + (lane >> 4)) * ld + row0 + (lane & 15);
    const double* Pm = S + (size_t)(k * TILE + (lane >> 4)) * ld + col0 + (lane & 15);
    const double* C = S + (size_t)(col0 + (lane >> 4)) * ld + row0 + (lane & 15);
    constexpr int NCH = TILE / (4 * KSTEPS);
    constexpr int NBUF = NCH > 1 ? 2 : 1;
    double av[NBUF][KSTEPS][SUB], bv[NBUF][KSTEPS][SUB];
    auto load_chunk = [&](int buf, int kc) {
#pragma unroll
        for (int s4 = 0; s4 < KSTEPS; ++s4) {
            const size_t off = (size_t)(kc + s4 * 4) * ld;
#pragma unroll
            for (int q = 0; q < SUB; ++q) { av[buf][s4][q] = Pm[off + q * 16]; bv[buf][s4][q] = Pn[off + q * 16]; }
        }
    };
    load_chunk(0, 0);
    double4_t acc[SUB][SUB], cv[SUB][SUB];
#pragma unroll
    for (int a = 0; a < SUB; ++a)
#pragma unroll
        for (int b = 0; b < SUB; ++b) acc[a][b] = (double4_t){ 0, 0, 0, 0 };
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int buf = ch % NBUF;
        if (ch + 1 < NCH) load_chunk((ch + 1) % NBUF, (ch + 1) * 4 * KSTEPS);
        else {
            // last chunk: the C block streams in behind the final MFMAs instead of in front of the first ones
#pragma unroll
            for (int a = 0; a < SUB; ++a)
#pragma unroll
                for (int b = 0; b < SUB; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cv[a][b][r] = C[(size_t)(a * 16 + 4 * r) * ld + b * 16];
        }
#pragma unroll
        for (int s4 = 0; s4 < KSTEPS; ++s4)
#pragma unroll
            for (int a = 0; a < SUB; ++a)
#pragma unroll
                for (int b = 0; b < SUB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[buf][s4][a], bv[buf][s4][b], acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < SUB; ++a)
#pragma unroll
        for (int b = 0; b < SUB; ++b) out[a][b] = cv[a][b] + acc[a][b];
}

// Rectangular form: (16 SUBM) columns x (16 SUBN) rows.  C_FIRST: the accumulators START as the C block (loaded before the first
// operand chunk; no second register set for C); otherwise they start at zero and C is added at the end, loaded behind the last
// chunk (below) -- also within the 256 registers that two wavefronts per SIMD leave each other.
template <int SUBM, int SUBN, int KSTEPS, bool C_FIRST, int NBUF>
__device__ __forceinline__ void update_rect(const double* __restrict__ S, int ld, int k, int row0, int col0, int lane, double4_t (&out)[SUBM][SUBN])
{
    const double* Pn = S + (size_t)(k * TILE + (lane >> 4)) * ld + row0 + (lane & 15);
    const double* Pm = S + (size_t)(k * TILE + (lane >> 4)) * ld + col0 + (lane & 15);
    const double* C = S + (size_t)(col0 + (lane >> 4)) * ld + row0 + (lane & 15);
#ifdef CHOL_RECT_KTOTAL      // TIMING PROBE ONLY (wrong results): the half-tile update over a panel of this many columns instead of 128 --
    constexpr int NCH = CHOL_RECT_KTOTAL / (4 * KSTEPS);      // what a launch that applies two panel columns at once would cost (profiles/HISTORY.md)
#else
    constexpr int NCH = TILE / (4 * KSTEPS);
#endif
    // NBUF operand buffers of KSTEPS panel columns x 4: NBUF - 1 chunks are in flight while one is multiplied (the loads return in
    // order, so the wait before chunk ch leaves the later ones outstanding)
    double av[NBUF][KSTEPS][SUBM], bv[NBUF][KSTEPS][SUBN];
    auto load_chunk = [&](int buf, int kc) {
#pragma unroll
        for (int s4 = 0; s4 < KSTEPS; ++s4) {
            const size_t off = (size_t)(kc + s4 * 4) * ld;
#pragma unroll
            for (int q = 0; q < SUBM; ++q) av[buf][s4][q] = Pm[off + q * 16];
#pragma unroll
            for (int q = 0; q < SUBN; ++q) bv[buf][s4][q] = Pn[off + q * 16];
        }
    };
#pragma unroll
    for (int p = 0; p < NBUF - 1; ++p) load_chunk(p, p * 4 * KSTEPS);
#pragma unroll
    for (int a = 0; a < SUBM; ++a)
#pragma unroll
        for (int b = 0; b < SUBN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[a][b][r] = C_FIRST ? C[(size_t)(a * 16 + 4 * r) * ld + b * 16] : 0.0;
    double4_t cv[SUBM][SUBN];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int buf = ch % NBUF;
        if (ch + NBUF - 1 < NCH) load_chunk((ch + NBUF - 1) % NBUF, (ch + NBUF - 1) * 4 * KSTEPS);
        if (!C_FIRST && ch == NCH - 1) {
            // the C block streams in behind the final chunk's products, into the registers the operand ring no longer needs (there is
            // no chunk left to prefetch): in front of the first products its latency -- C comes from memory, once per launch, while the
            // operands come from L2 -- was exposed in every task (2.72 -> 2.62 ms per factorisation at 6016)
#pragma unroll
            for (int a = 0; a < SUBM; ++a)
#pragma unroll
                for (int b = 0; b < SUBN; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) cv[a][b][r] = C[(size_t)(a * 16 + 4 * r) * ld + b * 16];
        }
#pragma unroll
        for (int s4 = 0; s4 < KSTEPS; ++s4)
#pragma unroll
            for (int a = 0; a < SUBM; ++a)
#pragma unroll
                for (int b = 0; b < SUBN; ++b)
                    out[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[buf][s4][a], bv[buf][s4][b], out[a][b], 0, 0, 0);
    }
    if (!C_FIRST) {
#pragma unroll
        for (int a = 0; a < SUBM; ++a)
#pragma unroll
            for (int b = 0; b < SUBN; ++b) out[a][b] = cv[a][b] + out[a][b];
    }
}

// NOT THE DEFAULT (MAGE_CHOL_BULK2_STAGED=1 selects it): measured 3.01 ms per factorisation against 2.87 for the form that reads its
// operands per wavefront -- eight barriers per task and the LDS round trip cost more than the halved L2 traffic returns.
// Half tile (128 rows x 64 columns) by one workgroup with the panel operands STAGED THROUGH LDS: per 16 panel columns the 128 + 64
// operand rows are fetched once by the workgroup (six 128-bit loads per thread, in flight while the previous chunk is multiplied)
// instead of once per wavefront -- a 64 x 32 wavefront tile needs 0.19 operand bytes per flop from L2, which two wavefronts per SIMD
// on 256 compute units cannot be fed; through LDS it is 0.094 from L2.  LDS: 2 buffers x 16 x 208 doubles (pitch 208: the four
// k rows of a fragment read land on alternating bank halves) = 52 KB of the 78 KB every workgroup of the launch owns anyway.
// operand pipeline of the half-tile update (update_rect): panel columns per chunk / 4, and chunks in the ring
#ifndef CHOL_RECT_KSTEPS
#define CHOL_RECT_KSTEPS 4
#endif
#ifndef CHOL_RECT_NBUF
#define CHOL_RECT_NBUF 2
#endif
constexpr int ST_KC = 16, ST_PITCH = 208;
// piece u of a thread: panel column kk = e / 96 of the chunk, 128-bit piece w = e % 96 of its 192 operand rows (e = 256 u + tid)
__device__ __forceinline__ const double2* st_src(const double* __restrict__ Pn, const double* __restrict__ Pm, int ld, int ch, int tid, int u)
{
    const int e = u * 256 + tid, kk = e / 96, w = e - kk * 96;
    return reinterpret_cast<const double2*>((w < 64 ? Pn + 2 * w : Pm + 2 * (w - 64)) + (size_t)(ch * ST_KC + kk) * ld);
}
__device__ __forceinline__ double2* st_dst(double* __restrict__ buf, int tid, int u)
{
    const int e = u * 256 + tid, kk = e / 96, w = e - kk * 96;
    return reinterpret_cast<double2*>(buf + kk * ST_PITCH + 2 * w);
}
#define ST_FETCH(ch) do { s0 = *st_src(Pn, Pm, ld, ch, tid, 0); s1 = *st_src(Pn, Pm, ld, ch, tid, 1); s2 = *st_src(Pn, Pm, ld, ch, tid, 2); \
                          s3 = *st_src(Pn, Pm, ld, ch, tid, 3); s4 = *st_src(Pn, Pm, ld, ch, tid, 4); s5 = *st_src(Pn, Pm, ld, ch, tid, 5); } while (0)
#define ST_PARK(buf) do { *st_dst(buf, tid, 0) = s0; *st_dst(buf, tid, 1) = s1; *st_dst(buf, tid, 2) = s2; \
                          *st_dst(buf, tid, 3) = s3; *st_dst(buf, tid, 4) = s4; *st_dst(buf, tid, 5) = s5; } while (0)
__device__ __forceinline__ void update_half_tile_staged(double* __restrict__ S, int ld, int k, int R0, int C0, double* __restrict__ sm, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const double* Pn = S + (size_t)(k * TILE) * ld + R0;       // 128 operand rows (the tile's rows) of panel column kk at Pn[kk * ld + row]
    const double* Pm = S + (size_t)(k * TILE) * ld + C0;       // 64 operand rows (the tile's columns)
    double2 s0, s1, s2, s3, s4, s5;         // (scalars: as an array the six pieces were kept in scratch memory)
    ST_FETCH(0);
    const int row0 = (wave & 1) * 64, col0 = (wave >> 1) * 32;
    double* C = S + (size_t)(C0 + col0 + (lane >> 4)) * ld + R0 + row0 + (lane & 15);
    double4_t acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = C[(size_t)(a * 16 + 4 * r) * ld + b * 16];
    ST_PARK(sm);
    __syncthreads();
    constexpr int NCH = TILE / ST_KC;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        const double* buf = sm + (ch & 1) * ST_KC * ST_PITCH;
        if (ch + 1 < NCH) ST_FETCH(ch + 1);
        const double* fb = buf + (lane >> 4) * ST_PITCH + (lane & 15);
#pragma unroll
        for (int s4 = 0; s4 < ST_KC / 4; ++s4) {
            double av[2], bv[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) av[a] = -fb[s4 * 4 * ST_PITCH + 128 + col0 + a * 16];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = fb[s4 * 4 * ST_PITCH + row0 + b * 16];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
        if (ch + 1 < NCH) ST_PARK(sm + ((ch + 1) & 1) * ST_KC * ST_PITCH);
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(size_t)(a * 16 + 4 * r) * ld + b * 16] = acc[a][b][r];
}

// Round 4: the half tile with its panel operands brought in by the LOAD-TO-LDS path (global_load_lds, 16 bytes per lane, no registers, no
// ds_write pass), double-buffered in 16-column chunks behind RAW barriers (s_barrier + lgkmcnt only: __syncthreads() would also wait for
// the loads in flight) -- the form the guide's GEMM recipes use.  What it is for: in the default form every WAVEFRONT fetches its own
// 64 + 32 operand rows from L2 (0.19 bytes per flop; ~14 TB/s of L2 -> CU traffic at the measured 53 % matrix-core utilisation), here a
// WORKGROUP fetches its 128 + 64 rows once (half of that) and the four wavefronts read fragments from LDS.  One instruction moves one
// panel column's 128 rows (1 KB, lanes along rows); the 64 column-side rows take the lower half of a wavefront.  LDS image per chunk:
// [16][GL_PN] + [16][GL_PM] doubles, pitches 144 / 80 so that the four panel columns of a fragment read land 32 banks apart.
// Sums in the same order as update_rect (k ascending, C added at the end): bit-identical.  MAGE_CHOL_BULK2_FORM=glds selects it.
// MEASURED SLOWER, like round 2's plain-load staging: 2.86 ms per factorisation against 2.565, every update-bound launch ~12 us longer
// (profiles/r04_chol_links.txt) -- the LDS round trip and a barrier per 16 columns cost more than the halved L2 traffic returns.  Not the default.
constexpr int GL_KC = 16, GL_PN = 144, GL_PM = 80;
constexpr int GL_BUF = GL_KC * (GL_PN + GL_PM);           // 3584 doubles = 28 KB per buffer, two buffers
__device__ __forceinline__ void glds16(const double* g, double* l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ void update_half_tile_glds(double* __restrict__ S, int ld, int k, int R0, int C0, double* __restrict__ sm, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const double* Pn = S + (size_t)(k * TILE) * ld + R0 + lane * 2;       // panel column kk of the tile's 128 rows: + kk * ld
    const double* Pm = S + (size_t)(k * TILE) * ld + C0 + lane * 2;       // ... of the tile's 64 columns (lanes 0-31)
    auto issue = [&](int ch, double* buf) {                                // wavefront w brings in panel columns w, w + 4, w + 8, w + 12 of the chunk
#pragma unroll
        for (int q = 0; q < GL_KC / 4; ++q) {
            const int kk = wave + 4 * q;
            glds16(Pn + (size_t)(ch * GL_KC + kk) * ld, buf + kk * GL_PN);
            if (lane < 32) glds16(Pm + (size_t)(ch * GL_KC + kk) * ld, buf + GL_KC * GL_PN + kk * GL_PM);
        }
    };
    const int row0 = (wave & 1) * 64, col0 = (wave >> 1) * 32;
    double* C = S + (size_t)(C0 + col0 + (lane >> 4)) * ld + R0 + row0 + (lane & 15);
    double4_t acc[2][4], cv[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){ 0, 0, 0, 0 };
    issue(0, sm);
    constexpr int NCH = TILE / GL_KC;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        double* buf = sm + (ch & 1) * GL_BUF;
        // my pieces of chunk ch have landed; behind the barrier everybody's have, and everybody has finished reading the other buffer
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (ch + 1 < NCH) issue(ch + 1, sm + ((ch + 1) & 1) * GL_BUF);
        const double* fn = buf + (lane >> 4) * GL_PN + row0 + (lane & 15);
        const double* fm = buf + GL_KC * GL_PN + (lane >> 4) * GL_PM + col0 + (lane & 15);
#pragma unroll
        for (int s4 = 0; s4 < GL_KC / 4; ++s4) {
            double av[2], bv[4];
#pragma unroll
            for (int a = 0; a < 2; ++a) av[a] = -fm[s4 * 4 * GL_PM + a * 16];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = fn[s4 * 4 * GL_PN + b * 16];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
        }
    }
    // C comes from memory once per launch: fetched behind the last chunk's products, added at the end (as update_rect does)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) cv[a][b][r] = C[(size_t)(a * 16 + 4 * r) * ld + b * 16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(size_t)(a * 16 + 4 * r) * ld + b * 16] = cv[a][b][r] + acc[a][b][r];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // nobody still reads the buffers when the workgroup's next use of LDS begins (none: it ends here)
}

// Grid of the trailing update of step k (tiles (i, j), j0 = k + 1 <= j <= i < nt), in dispatch order:
//   blocks 0..8   the NEXT diagonal tile (j0, j0): its 36 lower 16x16 blocks, one per wavefront (32 dependent MFMAs
//                 each instead of one wavefront grinding through a 64x64 quadrant: the tile is on the critical path).
//                 All write to S; blocks 1..8 then count up `flag` (agent-scope release), block 0 waits for 8 (acquire),
//                 pulls the tile into LDS and factors it in place -- the sequential diagonal factorisation of step
//                 k + 1 overlaps the rest of this update.
//   whole tiles   one 128x128 tile per block (tile index 1..), 64x64 per wavefront.
//   quarter tiles when the last round of whole tiles would occupy at most half of the compute units (n_q4 tiles), those
//                 tiles are cut into four 64x64 blocks, 32x32 per wavefront, so the round ends in a quarter of the time.
//   last m blocks rhs update y_i -= L_ik y_k.
std::atomic<bool> g_merge_disabled{ false };   // see chol_factor_solve

// development only (tools/chol_test.hip, CHOL_DBG_COL=k): time stamps of the workgroups of ONE chain-bound launch
__device__ long long g_syrk_dbg[32];
__device__ __forceinline__ void dbg_set(int dbg, int slot) { if (dbg && threadIdx.x == 0) g_syrk_dbg[slot] = wall_clock64(); }
__device__ __forceinline__ void dbg_max(int dbg, int slot) { if (dbg && threadIdx.x == 0) atomicMax((unsigned long long*)&g_syrk_dbg[slot], (unsigned long long)wall_clock64()); }
__device__ __forceinline__ void dbg_min(int dbg, int slot) { if (dbg && threadIdx.x == 0) atomicMin((unsigned long long*)&g_syrk_dbg[slot], (unsigned long long)wall_clock64()); }

constexpr int NDIAG = 9;           // workgroups on the next diagonal tile: 36 lower 16x16 blocks / 4 wavefronts

__host__ __device__ inline int syrk_quartered_tiles(int n_tiles /* incl. the diagonal one */, int n_cu)
{
    const int whole = n_tiles - 1;
    const int rem = whole % n_cu;
    return (rem > 0 && rem * 2 <= n_cu) ? rem : 0;
}

__device__ __forceinline__ void tile_of_index(int t, int& rt, int& ct)
{
    rt = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((rt + 1) * (rt + 2) / 2 <= t) ++rt;
    while (rt * (rt + 1) / 2 > t) --rt;
    ct = t - rt * (rt + 1) / 2;
}

// Tiles of the lower triangle (mt tile rows) in BAND order: bands of SYRK_BAND tile rows, inside a band column by column (so that
// consecutive tiles share their column block and all tiles of a band share its few row blocks).  u = 0 is tile (0, 0).
constexpr int SYRK_BAND = 4;
__device__ __forceinline__ void tile_of_band_order(int u, int mt, int& rt, int& ct)
{
    int r0, c0;
    tile_of_index(u, r0, c0);                        // r0 = the row whose row-major run holds u: bands start at row boundaries
    const int R0 = (r0 / SYRK_BAND) * SYRK_BAND;
    const int h = min(SYRK_BAND, mt - R0);
    int v = u - R0 * (R0 + 1) / 2;
    const int full = (R0 + 1) * h;                   // columns 0 .. R0 carry all h rows of the band
    if (v < full) { ct = v / h; rt = R0 + v % h; return; }
    v -= full;
    rt = R0 + h - 1; ct = rt;
    for (int d = 1; d < h; ++d) {                    // columns R0 + d: rows R0 + d .. R0 + h - 1
        const int cnt = h - d;
        if (v < cnt) { ct = R0 + d; rt = R0 + d + v; return; }
        v -= cnt;
    }
}

// ---- panel solve merged into the trailing update's launch (flag[0]: arrivals at the split diagonal tile, flag[1]: the last tile
// column whose diagonal tile is factored and in memory, flag[2]: parts of first-column tiles written, cumulative over the launches of a
// factorisation; k_trsm_panel of column 0 zeroes all three).  Producers: every store of the workgroup done, ONE agent-scope release
// (an L2 write-back), then the count.  Consumers (the strips, dispatched behind every producer they wait for): relaxed polls, one acquire.
__device__ __forceinline__ void publish_column_part(int* __restrict__ flag, int tid)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(flag + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void wait_for_column(int* __restrict__ flag, int j0, int col_target, double* __restrict__ stall, int lane)
{
    if (lane == 0) {
        int spins = 0;
        while ((__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < j0 ||
                __hip_atomic_load(flag + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < col_target) && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2);
        if (spins >= (1 << 22)) *stall = 2.0;           // (the value names the wait that ran out: 1 split diagonal tile, 2 merged panel solve, 3 backward solve)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __builtin_amdgcn_wave_barrier();
}

// The merged strip with the write-through hand-off (k_syrk_update<1, true>; the strip is ON the chain in the columns that use it).
// What it waits for arrives in two parts, and it no longer waits for both before touching either:
//   1. its own rows of column j0 -- written by the update's first-column workgroups with plain stores + release + count, complete long
//      before the tile is factored: poll flag[2], ONE acquire, fetch the rows;
//   2. L_j0j0 and the block inverses -- written THROUGH by workgroup 0 (sc1 stores, no release fence): poll flag[1], then agent-scope
//      loads (not served from this compute unit's L1, so no second acquire), all up front as in trsm_strip.
// Same operations in the same order as trsm_strip: bit-identical.
__device__ __forceinline__ bool poll_at_least(const int* __restrict__ word, int target, int lane)
{
    bool ok = true;
    if (lane == 0) {
        int spins = 0;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
        ok = spins < (1 << 22);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    return ok;
}
__device__ __forceinline__ void trsm_strip_wt(double* __restrict__ S, double* __restrict__ y, int ld, int k, int strip, bool is_rhs,
                                              const double* __restrict__ Linv_k, int* __restrict__ flag, int col_target, double* __restrict__ stall, int lane)
{
    double* base;
    size_t cstride;
    bool live;
    if (!is_rhs) {
        base = S + (size_t)(k * TILE) * ld + (size_t)(k + 1) * TILE + strip * NB + (lane & 15);
        cstride = (size_t)ld;
        live = true;
        if (!poll_at_least(flag + 2, col_target, lane) && lane == 0) *stall = 2.0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __builtin_amdgcn_wave_barrier();
    } else {
        base = y + (size_t)k * TILE;       // this workgroup's own row, just updated
        cstride = 1;
        live = (lane & 15) == 0;
    }
    double4_t Acc[NBLK];
#pragma unroll
    for (int c = 0; c < NBLK; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) Acc[c][r] = live ? base[(size_t)(c * NB + (lane >> 4) + 4 * r) * cstride] : 0.0;
    if (!poll_at_least(flag + 1, k, lane) && lane == 0) *stall = 2.0;
    const double* Lop = S + (size_t)(k * TILE + (lane >> 4)) * ld + (size_t)k * TILE + (lane & 15);
    const double* Lio = Linv_k + (lane & 15) * NB + (lane >> 4);
    double lop[NBLK][NBLK][4], lio[NBLK][4];
#pragma unroll
    for (int c = 0; c < NBLK; ++c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) lio[c][r] = __hip_atomic_load(Lio + c * NB * NB + 4 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < c; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) lop[c][j][r] = -__hip_atomic_load(Lop + (size_t)(j * NB + 4 * r) * ld + c * NB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double4_t Y[NBLK];
#pragma unroll
    for (int c = 0; c < NBLK; ++c) {
        double4_t acc = Acc[c];
#pragma unroll
        for (int j = 0; j < c; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(lop[c][j][r], Y[j][r], acc, 0, 0, 0);
        double4_t yc = { 0, 0, 0, 0 };
#pragma unroll
        for (int r = 0; r < 4; ++r) yc = __builtin_amdgcn_mfma_f64_16x16x4f64(lio[c][r], acc[r], yc, 0, 0, 0);
        Y[c] = yc;
        if (live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) base[(size_t)(c * NB + (lane >> 4) + 4 * r) * cstride] = yc[r];
        }
    }
}

// One 16-row strip by the FOUR wavefronts of its workgroup (round 4).  A strip is a chain of 176 matrix-core operations and every
// f64 MFMA holds its SIMD's pipe for 64 cycles, dependent or not: one wavefront needs 11.3 k cycles (4.7 us) whatever it overlaps.
// But only 8 of a step's operations are on the chain -- acc_{j+1} -= L(j+1, j) Y_j and Y_{j+1} = Linv_{j+1} acc_{j+1}; the updates
// of the later block columns are independent of each other.  Wavefront w owns block columns {w, 7 - w} (9 products each): the owner
// of column j forms Y_j, leaves it in LDS (accumulator layout = B operand of the updates), one LDS-only barrier, every wavefront
// applies Y_j to the columns it owns -- the next step's column first.  Operands per wavefront: 9 blocks instead of 36.
// Same operations in the same order per block column as trsm_strip / trsm_strip_wt: bit-identical.
// WT: L_kk and the block inverses come through agent-scope loads (write-through hand-off); WAIT: poll the launch's flags first
// (wavefront 0 polls, the others wait at the barrier).  ysh: (NBLK - 1) * 256 doubles of LDS.
template <bool WT, bool WAIT, int W>
__device__ __forceinline__ void trsm_strip_4w_body(double* __restrict__ base, size_t cstride, bool live, const double* __restrict__ S, int ld, int k,
                                                   const double* __restrict__ Linv_k, int* __restrict__ flag, double* __restrict__ stall, int lane,
                                                   double* __restrict__ ysh)
{
    constexpr int C0 = W, C1 = NBLK - 1 - W;
    double4_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        acc0[r] = live ? base[(size_t)(C0 * NB + (lane >> 4) + 4 * r) * cstride] : 0.0;
        acc1[r] = live ? base[(size_t)(C1 * NB + (lane >> 4) + 4 * r) * cstride] : 0.0;
    }
    if (WAIT) {
        if (W == 0 && !poll_at_least(flag + 1, k, lane) && lane == 0) *stall = 2.0;
        tile_barrier<true>();
    }
    const double* Lop = S + (size_t)(k * TILE + (lane >> 4)) * ld + (size_t)k * TILE + (lane & 15);
    const double* Lio = Linv_k + (lane & 15) * NB + (lane >> 4);
    auto ld1 = [](const double* p) -> double { return WT ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; };
    double lio0[4], lio1[4], lop0[C0 > 0 ? C0 : 1][4], lop1[C1][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { lio0[r] = ld1(Lio + C0 * NB * NB + 4 * r); lio1[r] = ld1(Lio + C1 * NB * NB + 4 * r); }
#pragma unroll
    for (int j = 0; j < C0; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) lop0[j][r] = -ld1(Lop + (size_t)(j * NB + 4 * r) * ld + C0 * NB);
#pragma unroll
    for (int j = 0; j < C1; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) lop1[j][r] = -ld1(Lop + (size_t)(j * NB + 4 * r) * ld + C1 * NB);
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
        double4_t Yj = { 0, 0, 0, 0 };
        const bool mine = j == C0 || j == C1;
        if (mine) {
            const double4_t a = j == C0 ? acc0 : acc1;
#pragma unroll
            for (int r = 0; r < 4; ++r) Yj = __builtin_amdgcn_mfma_f64_16x16x4f64(j == C0 ? lio0[r] : lio1[r], a[r], Yj, 0, 0, 0);
            if (j < NBLK - 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ysh[j * 256 + r * 64 + lane] = Yj[r];
            }
            if (live) {
#pragma unroll
                for (int r = 0; r < 4; ++r) base[(size_t)(j * NB + (lane >> 4) + 4 * r) * cstride] = Yj[r];
            }
        }
        if (j == NBLK - 1) break;
        tile_barrier<true>();
        if (!mine) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Yj[r] = ysh[j * 256 + r * 64 + lane];
        }
        // the column of the next step first
        if (C1 > j && C1 == j + 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lop1[j][r], Yj[r], acc1, 0, 0, 0);
        }
        if (C0 > j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(lop0[j][r], Yj[r], acc0, 0, 0, 0);
        }
        if (C1 > j && C1 != j + 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(lop1[j][r], Yj[r], acc1, 0, 0, 0);
        }
    }
}
template <bool WT, bool WAIT>
__device__ __forceinline__ void trsm_strip_4w(double* __restrict__ S, double* __restrict__ y, int ld, int k, int strip, bool is_rhs, const double* __restrict__ Linv_k,
                                              int* __restrict__ flag, int col_target, double* __restrict__ stall, int lane, int wave, double* __restrict__ ysh)
{
    double* base;
    size_t cstride;
    bool live;
    if (!is_rhs) {
        base = S + (size_t)(k * TILE) * ld + (size_t)(k + 1) * TILE + strip * NB + (lane & 15);
        cstride = (size_t)ld;
        live = true;
        if (WAIT) {
            if (wave == 0 && !poll_at_least(flag + 2, col_target, lane) && lane == 0) *stall = 2.0;
            tile_barrier<true>();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    } else {
        base = y + (size_t)k * TILE;       // this workgroup's own row (just updated when WAIT)
        cstride = 1;
        live = (lane & 15) == 0;
    }
    switch (wave) {
        case 0: trsm_strip_4w_body<WT, WAIT, 0>(base, cstride, live, S, ld, k, Linv_k, flag, stall, lane, ysh); break;
        case 1: trsm_strip_4w_body<WT, WAIT, 1>(base, cstride, live, S, ld, k, Linv_k, flag, stall, lane, ysh); break;
        case 2: trsm_strip_4w_body<WT, WAIT, 2>(base, cstride, live, S, ld, k, Linv_k, flag, stall, lane, ysh); break;
        default: trsm_strip_4w_body<WT, WAIT, 3>(base, cstride, live, S, ld, k, Linv_k, flag, stall, lane, ysh); break;
    }
}

// PHASED strip of the merged panel solve (round 4): the factoring workgroup publishes block column c of L_kk (and the block inverses)
// as it goes (potrf_tile_lds<.., 4>: write-through stores, progress words raised one in-tile iteration later, when the stores have long
// been acknowledged), and a strip works in three phases behind three polls of ONE word each instead of waiting for the whole tile:
//   columns 0-3 published  ->  steps 0-3 and the products of the later block columns with Y_0 .. Y_3   (26 of the 36 products)
//   columns 4-5 published  ->  steps 4, 5 and their products
//   tile factored (flag[1]) ->  steps 6, 7: two inverse products and one update behind the last fetch
// What is left on the chain behind the factorisation is one fetch, twelve matrix-core operations and a store.  Per block column the
// products meet the accumulator in the same order as in trsm_strip_wt: bit-identical.  (The first pipelined form polled the OPERANDS
// for a sentinel, two hundred strips re-reading L_kk past the L2: the polling took fabric bandwidth from the factoring workgroup.)
__device__ __forceinline__ bool poll_progress(const int* __restrict__ word, int target, int lane)
{
    bool ok = true;
    if (lane == 0) {
        int spins = 0;
        // (a pause between polls: up to two hundred strips watch this word while the factoring workgroup works through memory)
        // (pauses of 1 ... 32 between polls measured the same, 2.52-2.53 ms per factorisation: one word, one cache line)
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(8);
        ok = spins < (1 << 20);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    return ok;
}
__device__ __forceinline__ void trsm_strip_phased(double* __restrict__ S, double* __restrict__ y, int ld, int k, int strip, bool is_rhs,
                                                  const double* __restrict__ Linv_k, const double* __restrict__ Lpub, int* __restrict__ flag, int col_target,
                                                  double* __restrict__ stall, int lane)
{
    double* base;
    size_t cstride;
    bool live;
    if (!is_rhs) {
        base = S + (size_t)(k * TILE) * ld + (size_t)(k + 1) * TILE + strip * NB + (lane & 15);
        cstride = (size_t)ld;
        live = true;
        if (!poll_at_least(flag + 2, col_target, lane) && lane == 0) *stall = 2.0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __builtin_amdgcn_wave_barrier();
    } else {
        base = y + (size_t)k * TILE;       // this workgroup's own row, just updated
        cstride = 1;
        live = (lane & 15) == 0;
    }
    double4_t Acc[NBLK], Y[NBLK];
#pragma unroll
    for (int c = 0; c < NBLK; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) Acc[c][r] = live ? base[(size_t)(c * NB + (lane >> 4) + 4 * r) * cstride] : 0.0;
    // operand element (row = lane & 15, column = 4 r + (lane >> 4)) of a published block; of the row-major block inverse
    const double* Lop = Lpub + (lane >> 4) * NB + (lane & 15);
    const double* Lio = Linv_k + (lane & 15) * NB + (lane >> 4);
    auto ld_inv = [&](int c, double (&o)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = __hip_atomic_load(Lio + c * NB * NB + 4 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto ld_blk = [&](int c, int j, double (&o)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = -__hip_atomic_load(Lop + (size_t)(c * (c - 1) / 2 + j) * NB * NB + (size_t)r * 4 * NB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto step = [&](int c, const double (&inv)[4]) {          // Y_c = Linv_c Acc_c, stored
        double4_t yc = { 0, 0, 0, 0 };
#pragma unroll
        for (int r = 0; r < 4; ++r) yc = __builtin_amdgcn_mfma_f64_16x16x4f64(inv[r], Acc[c][r], yc, 0, 0, 0);
        Y[c] = yc;
        if (live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) base[(size_t)(c * NB + (lane >> 4) + 4 * r) * cstride] = yc[r];
        }
    };
    auto update = [&](int c, int j, const double (&l)[4]) {   // Acc_c -= L(c, j) Y_j
#pragma unroll
        for (int r = 0; r < 4; ++r) Acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(l[r], Y[j][r], Acc[c], 0, 0, 0);
    };
    const int pbase = 8 * k;
    // A strip that starts when the tile is already factored (the update-bound end of the merged columns: strip workgroups are the last of
    // the grid) has nothing to overlap: every operand in ONE round of loads, as trsm_strip_wt does, instead of three.
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= k) {
        double inv[NBLK][4], l[NBLK][NBLK][4];
#pragma unroll
        for (int c = 0; c < NBLK; ++c) {
            ld_inv(c, inv[c]);
#pragma unroll
            for (int j = 0; j < c; ++j) ld_blk(c, j, l[c][j]);
        }
#pragma unroll
        for (int c = 0; c < NBLK; ++c) {
#pragma unroll
            for (int j = 0; j < c; ++j) update(c, j, l[c][j]);
            step(c, inv[c]);
        }
        return;
    }
    // ---- phase 1: block columns 0 .. 3
    if (!poll_progress(flag + 4, pbase + 3, lane) && lane == 0) *stall = 2.0;
    {
        double inv[4][4], l[NBLK][4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) ld_inv(c, inv[c]);
#pragma unroll
        for (int c = 1; c < NBLK; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j < c) ld_blk(c, j, l[c][j]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int j = 0; j < c; ++j) update(c, j, l[c][j]);
            step(c, inv[c]);
        }
#pragma unroll
        for (int c = 4; c < NBLK; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) update(c, j, l[c][j]);
    }
    // ---- phase 2: block columns 4, 5
    if (!poll_progress(flag + 4, pbase + 5, lane) && lane == 0) *stall = 2.0;
    {
        double inv4[4], inv5[4], l54[4], l64[4], l74[4], l65[4], l75[4];
        ld_inv(4, inv4); ld_inv(5, inv5);
        ld_blk(5, 4, l54); ld_blk(6, 4, l64); ld_blk(7, 4, l74); ld_blk(6, 5, l65); ld_blk(7, 5, l75);
        step(4, inv4);
        update(5, 4, l54);
        step(5, inv5);
        update(6, 4, l64); update(6, 5, l65);
        update(7, 4, l74); update(7, 5, l75);
    }
    // ---- phase 3: the tile is factored
    if (!poll_at_least(flag + 1, k, lane) && lane == 0) *stall = 2.0;
    {
        double inv6[4], inv7[4], l76[4];
        ld_inv(6, inv6); ld_inv(7, inv7); ld_blk(7, 6, l76);
        step(6, inv6);
        update(7, 6, l76);
        step(7, inv7);
    }
}

// the panel solve of tile column k as a launch of its own (the update-bound columns; column 0)
// (two kernels, not one with a run-time choice: the one-wavefront strip keeps all 36 operand blocks in ~380 registers, and at that
// size only ONE four-wavefront workgroup fits a compute unit -- the 369 strips of column 0 would take two rounds)
__device__ __forceinline__ void panel_launch_duties(double* __restrict__ Linv_k, int k, int nt, int* __restrict__ queue, int queue_start, int tid, int nthreads)
{
    if (blockIdx.x == 0 && tid == 0) { queue[0] = 0; if (k == 0) { queue[1] = 0; queue[2] = 0; queue[3] = 0; queue[4] = 0; queue[5] = 0; queue[6] = 0; } }     // hand-off counters of the launches that follow (queue_start < 0: also fill the pipelined strips' sentinels)
    if (k == 0 && queue_start < 0) {
        // what the pipelined panel solves of the later columns poll: the block inverses of tiles 1 .. nt - 1 and every tile's scratch
        // blocks start as the sentinel (this launch is the second of a factorisation; those areas are first written 19+ launches later)
        unsigned long long* W = reinterpret_cast<unsigned long long*>(Linv_k);
        const size_t first = (size_t)NBLK * NB * NB, total = (size_t)nt * (NBLK * NB * NB + LPUB_TILE_DOUBLES);
        for (size_t i = first + (size_t)blockIdx.x * nthreads + tid; i < total; i += (size_t)gridDim.x * nthreads) W[i] = X_SENTINEL;
    }
}
// one strip per workgroup, by its four wavefronts (trsm_strip_4w; MAGE_CHOL_STRIP_4W=1)
__global__ __launch_bounds__(256, 2) void k_trsm_panel(double* __restrict__ S, double* __restrict__ y, int ld, int k, int nt,
                                                       const double* __restrict__ Linv_k, int* __restrict__ queue, int queue_start)
{
    __shared__ double ysh[(NBLK - 1) * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    panel_launch_duties(const_cast<double*>(Linv_k), k, nt, queue, queue_start, tid, 256);
    const int n_strips = (nt - k - 1) * NBLK;
    trsm_strip_4w<false, false>(S, y, ld, k, (int)blockIdx.x, (int)blockIdx.x == n_strips, Linv_k, nullptr, 0, nullptr, lane, wave, ysh);
}
// the same by one wavefront (the default)
__global__ __launch_bounds__(64) void k_trsm_panel_1w(double* __restrict__ S, double* __restrict__ y, int ld, int k, int nt,
                                                      const double* __restrict__ Linv_k, int* __restrict__ queue, int queue_start)
{
    const int lane = threadIdx.x;
    panel_launch_duties(const_cast<double*>(Linv_k), k, nt, queue, queue_start, lane, 64);
    const int n_strips = (nt - k - 1) * NBLK;
    trsm_strip(S, y, ld, k, (int)blockIdx.x, (int)blockIdx.x == n_strips, Linv_k, lane);
}

// The merged strip as a PRODUCT (round 4): with the tile's full inverse in memory (potrf_tile_lds<.., 3>) the strip is
//     X = A L^-T,   i.e. on the transposed unknown   Y_c = sum_{j <= c} Linv[c][j] A_j^T        (c = 0 .. 7, 16 columns each)
// and nothing depends on anything: the FOUR wavefronts of the strip's workgroup take the block columns {w, 7 - w} -- 36 matrix-core
// operations each instead of one wavefront's chain of 176 (every f64 MFMA occupies its SIMD's pipe for 64 cycles, dependent or not:
// the substitution form is one SIMD's 5.4 us).  Waits as in trsm_strip_wt: the rows first (count, acquire, fetch), then the inverse
// (flag, agent-scope loads).  INVg: block (c, j) at (c (c + 1) / 2 + j) * 256, column-major inside the block.
__device__ __forceinline__ void trsm_strip_gemm(double* __restrict__ S, double* __restrict__ y, int ld, int k, int strip, bool is_rhs,
                                                const double* __restrict__ INVg, int* __restrict__ flag, int col_target, double* __restrict__ stall, int lane, int wave)
{
    double* base;
    size_t cstride;
    bool live;
    if (!is_rhs) {
        base = S + (size_t)(k * TILE) * ld + (size_t)(k + 1) * TILE + strip * NB + (lane & 15);
        cstride = (size_t)ld;
        live = true;
        if (!poll_at_least(flag + 2, col_target, lane) && lane == 0) *stall = 2.0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __builtin_amdgcn_wave_barrier();
    } else {
        base = y + (size_t)k * TILE;       // this workgroup's own row, just updated
        cstride = 1;
        live = (lane & 15) == 0;
    }
    const int c1 = wave, c2 = NBLK - 1 - wave;            // c1 <= 3 < c2
    double Af[NBLK][4];                                   // A_j^T fragments, j <= c2
#pragma unroll
    for (int j = 0; j < NBLK; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) Af[j][r] = (live && j <= c2) ? base[(size_t)(j * NB + (lane >> 4) + 4 * r) * cstride] : 0.0;
    if (!poll_at_least(flag + 1, k, lane) && lane == 0) *stall = 2.0;
    const double* Ig = INVg + (lane >> 4) * NB + (lane & 15);      // fragment r of a block: + 64 r
    double4_t y1 = { 0, 0, 0, 0 }, y2 = { 0, 0, 0, 0 };
    // the operands of both columns up front (36 blocks' fragments at most: c1 + 1 + c2 + 1 = 9 blocks, 36 loads)
    double I1[4][4], I2[NBLK][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) I1[j][r] = j <= c1 ? __hip_atomic_load(Ig + (size_t)(c1 * (c1 + 1) / 2 + j) * NB * NB + 64 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
    for (int j = 0; j < NBLK; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) I2[j][r] = j <= c2 ? __hip_atomic_load(Ig + (size_t)(c2 * (c2 + 1) / 2 + j) * NB * NB + 64 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
        if (j <= c2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(I2[j][r], Af[j][r], y2, 0, 0, 0);
        }
        if (j < 4 && j <= c1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(I1[j][r], Af[j][r], y1, 0, 0, 0);
        }
    }
    if (live) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            base[(size_t)(c1 * NB + (lane >> 4) + 4 * r) * cstride] = y1[r];
            base[(size_t)(c2 * NB + (lane >> 4) + 4 * r) * cstride] = y2[r];
        }
    }
}

// MERGE is a template parameter (0 no panel solve in the launch, 1 merged strips, 2 pipelined strips): with the three forms in one
// body the strip roles -- ~420 live registers each -- were allocated against each other and 290 registers went to scratch memory,
// some of it between the matrix-core operations of the strip that sits on the chain.
// WT (with MERGE == 1): the write-through form of the two hand-offs that sit on the chain (the split diagonal tile -> workgroup 0, and
// workgroup 0 -> the strips): sc1 stores and loads instead of release / acquire fences (load_tile_packed_wt above).
// GEMM (with WT): workgroup 0 also builds the factored tile's full inverse and the strips are products on four wavefronts (trsm_strip_gemm).
template <int MERGE, bool WT = false, bool GEMM = false>
__global__ __launch_bounds__(256) void k_syrk_update(double* __restrict__ S, double* __restrict__ y, int ld, int k, int nt,
                                                     double* __restrict__ Linv_next, double* __restrict__ ok, double* __restrict__ stall, int* __restrict__ flag, int n_q4, int col_target,
                                                     int dbg, double* __restrict__ Lpub_next, int strip4w)
{
    constexpr int merge = MERGE;
    extern __shared__ double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0 = k + 1, mt = nt - j0;
    const int n_tiles = mt * (mt + 1) / 2;
    const int n_whole = n_tiles - 1 - n_q4;                       // tile indices 1 .. n_whole
    const int first_q4 = NDIAG + n_whole, first_rhs = first_q4 + 4 * n_q4;
    const int bid = blockIdx.x;
    if (bid >= first_rhs + mt) {
        // ---- merged panel solve of tile column j0 (merge != 0): one strip per workgroup (wavefront 0; all four in the product form), behind everything it waits for
        if constexpr (GEMM) {
            dbg_min(dbg, 6);
            trsm_strip_gemm(S, y, ld, j0, bid - (first_rhs + mt), false, Lpub_next, flag, col_target, stall, lane, wave);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg_max(dbg, 8);
            return;
        }
        if constexpr (MERGE == 1 && WT && !GEMM) {
            if (strip4w) {          // the strip by all four wavefronts (trsm_strip_4w); strip4w = 0: wavefront 0 alone, as below
                dbg_min(dbg, 6);
                trsm_strip_4w<true, true>(S, y, ld, j0, bid - (first_rhs + mt), false, Linv_next, flag, col_target, stall, lane, wave, sm);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (dbg) { __syncthreads(); dbg_max(dbg, 8); }
                return;
            }
        }
        if (wave != 0) return;
        dbg_min(dbg, 6);
        if constexpr (MERGE == 3) {           // phased against the factorisation of L_j0j0 (potrf_tile_lds<.., 4>)
            trsm_strip_phased(S, y, ld, j0, bid - (first_rhs + mt), false, Linv_next, Lpub_next, flag, col_target, stall, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg_max(dbg, 8);
            return;
        }
        if constexpr (MERGE == 0) return;
        else if constexpr (MERGE == 2) {      // pipelined against the factorisation of L_j0j0 (TilePublish): only the first-column tiles must be complete
            wait_for_column(flag, 0, col_target, stall, lane);
            dbg_max(dbg, 7);
            trsm_strip_pipelined(S, y, ld, j0, bid - (first_rhs + mt), false, Linv_next, Lpub_next, stall, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg_max(dbg, 8);
            return;
        } else if constexpr (WT) {
            trsm_strip_wt(S, y, ld, j0, bid - (first_rhs + mt), false, Linv_next, flag, col_target, stall, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg_max(dbg, 8);
            return;
        } else {
            wait_for_column(flag, j0, col_target, stall, lane);
            dbg_max(dbg, 7);
            trsm_strip(S, y, ld, j0, bid - (first_rhs + mt), false, Linv_next, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg_max(dbg, 8);
            return;
        }
    }
    if (bid >= first_rhs) {
        const int i = k + 1 + (bid - first_rhs);
        if (i >= nt) return;
        const double* Lik = S + (size_t)(k * TILE) * ld + (size_t)i * TILE;
        const double* yk = y + (size_t)k * TILE;
        for (int r = tid; r < TILE; r += 256) {
            double acc = 0;
#pragma unroll 8
            for (int c = 0; c < TILE; ++c) acc = __builtin_fma(Lik[(size_t)c * ld + r], yk[c], acc);
            y[(size_t)i * TILE + r] -= acc;
        }
        if constexpr (MERGE != 0) {
            if (i == j0) {
                // the rhs row of the merged panel solve: y_j0 is this workgroup's own (just updated), L_j0j0 comes from workgroup 0
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if constexpr (GEMM) { trsm_strip_gemm(S, y, ld, j0, 0, true, Lpub_next, flag, 0, stall, lane, wave); return; }
                if constexpr (MERGE == 1 && WT && !GEMM) {
                    if (strip4w) { trsm_strip_4w<true, true>(S, y, ld, j0, 0, true, Linv_next, flag, 0, stall, lane, wave, sm); return; }
                }
                if (wave != 0) return;
                if constexpr (MERGE == 3) trsm_strip_phased(S, y, ld, j0, 0, true, Linv_next, Lpub_next, flag, 0, stall, lane);
                else if constexpr (MERGE == 2) trsm_strip_pipelined(S, y, ld, j0, 0, true, Linv_next, Lpub_next, stall, lane);
                else if constexpr (WT) trsm_strip_wt(S, y, ld, j0, 0, true, Linv_next, flag, 0, stall, lane);
                else {
                    wait_for_column(flag, j0, 0, stall, lane);
                    trsm_strip(S, y, ld, j0, 0, true, Linv_next, lane);
                }
            }
        }
        return;
    }
    if (bid < NDIAG) {
        if (bid == 0) dbg_set(dbg, 0);
        // 16x16 block u = 4 bid + wave of the lower triangle of the diagonal tile, (bi, bj), bi >= bj
        const int u = bid * 4 + wave;
        int bi, bj;
        tile_of_index(u, bi, bj);
        const int row0 = j0 * TILE + bi * NB, col0 = j0 * TILE + bj * NB;
        double4_t out[1][1];
        update_block<1, 32>(S, ld, k, row0, col0, lane, out);
        double* C = S + (size_t)(col0 + (lane >> 4)) * ld + row0 + (lane & 15);
        if constexpr (WT) {
            // written THROUGH (8-byte agent-scope stores: the accumulator layout offers nothing wider): no release fence below
#pragma unroll
            for (int r = 0; r < 4; ++r) __hip_atomic_store(C + (size_t)(4 * r) * ld, out[0][0][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(size_t)(4 * r) * ld] = out[0][0][r];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (bid != 0) {
            // publish: all stores of the workgroup done -> (agent-scope release, unless they went through) -> count
            if (tid == 0) {
                if constexpr (!WT) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        // block 0: wait for the eight others (bounded spin; relaxed polls, one acquire unless the tile comes through sc1 loads), pull the tile, factor it
        dbg_set(dbg, 1);
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NDIAG - 1 && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(1);
            if (spins >= (1 << 24)) *stall = 1.0;        // a producer never arrived: reported as a device error (never folded into "not positive definite")
            if constexpr (!WT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (merge) __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // no panel-solve launch follows to reset it
        }
        __syncthreads();
        dbg_set(dbg, 2);
        double* A = sm;
        double* T = S + (size_t)(j0 * TILE) * ld + (size_t)j0 * TILE;
        if constexpr (WT) load_tile_packed_wt(A, T, ld, tid);
        else load_tile_packed(A, T, ld, tid);
        __syncthreads();
        dbg_set(dbg, 3);
        bool failed;
        if constexpr (MERGE == 3) failed = potrf_tile_lds<false, LayPacked, 4>(A, sm + PACKED_TILE_DOUBLES, Linv_next, tid, NBLK, TilePublish{ Lpub_next, nullptr, flag + 4, 8 * j0 });
        else if constexpr (MERGE == 2) failed = potrf_tile_lds<false, LayPacked, 2>(A, sm + PACKED_TILE_DOUBLES, Linv_next, tid, NBLK, TilePublish{ Lpub_next });
        else if constexpr (GEMM) failed = potrf_tile_lds<false, LayPacked, 3>(A, sm + PACKED_TILE_DOUBLES, Linv_next, tid, NBLK,
                                                                              TilePublish{ nullptr, Lpub_next });
        else if constexpr (WT) failed = potrf_tile_lds<false, LayPacked, 1>(A, sm + PACKED_TILE_DOUBLES, Linv_next, tid);
        else failed = potrf_tile_lds<false, LayPacked>(A, sm + PACKED_TILE_DOUBLES, Linv_next, tid);
        dbg_set(dbg, 4);
        if constexpr (MERGE == 3) {
            // the strips read the published blocks and the block inverses, all written through: raise the flag FIRST, the factor itself goes to
            // S afterwards (the backward solve reads it there, launches later)
            if (tid == 0 && failed) *ok = 0.0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flag + 1, j0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dbg_set(dbg, 5);
            store_tile_packed(T, A, ld, tid);
            return;
        }
        if constexpr (GEMM) {
            // the strips read only the inverse (written through by the factorisation): publish FIRST, store the factor itself afterwards
            if (tid == 0 && failed) *ok = 0.0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flag + 1, j0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            dbg_set(dbg, 5);
            store_tile_packed(T, A, ld, tid);
            return;
        }
        if constexpr (WT) store_tile_packed_wt(T, A, ld, tid);
        else store_tile_packed(T, A, ld, tid);
        if (tid == 0 && failed) *ok = 0.0;
        if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg_set(dbg, 5); }
        if (merge == 1) {      // L_j0j0 and its block inverses are in memory: the strips of this launch may start (pipelined strips need no release: write-through)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                if constexpr (!WT) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __hip_atomic_store(flag + 1, j0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    if (bid >= first_q4) {
        // quarter of tile index n_whole + 1 + (bid - first_q4) / 4
        const int q = bid - first_q4;
        int rt, ct;
        tile_of_index(n_whole + 1 + (q >> 2), rt, ct);
        const int row0 = (j0 + rt) * TILE + ((q >> 0) & 1) * 64 + (wave & 1) * 32;
        const int col0 = (j0 + ct) * TILE + ((q >> 1) & 1) * 64 + (wave >> 1) * 32;
        double4_t out[2][2];
        update_block<2, 8>(S, ld, k, row0, col0, lane, out);
        double* C = S + (size_t)(col0 + (lane >> 4)) * ld + row0 + (lane & 15);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) C[(size_t)(a * 16 + 4 * r) * ld + b * 16] = out[a][b][r];
        if (merge && ct == 0) publish_column_part(flag, tid);
        return;
    }
    int rt, ct;
    tile_of_index(bid - NDIAG + 1, rt, ct);
    const int row0 = (j0 + rt) * TILE + (wave & 1) * 64, col0 = (j0 + ct) * TILE + (wave >> 1) * 64;
    double4_t out[4][4];
    update_block<4, 8>(S, ld, k, row0, col0, lane, out);
    double* C = S + (size_t)(col0 + (lane >> 4)) * ld + row0 + (lane & 15);
    if (strip4w & 2) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) store_through(C + (size_t)(a * 16 + 4 * r) * ld + b * 16, out[a][b][r]);
    } else {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[(size_t)(a * 16 + 4 * r) * ld + b * 16] = out[a][b][r];
    }
    if (merge && ct == 0) publish_column_part(flag, tid);
    if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg_max(dbg, ct == 0 ? 10 : 9); }
}

// (Round 4 tried the panel solve of column k + 1 INSIDE this launch too -- first-column tiles first, strip workgroups in the middle of
// the grid, where they wait for nobody: bit-identical and 50-70 us SLOWER per factorisation.  What the separate ~11 us panel-solve
// launch costs is mostly the write-back of the update's dirty tiles at the kernel boundary, which the next launch then pays instead;
// profiles/r04_chol_merged_halftile_rejected.txt has the numbers and the per-launch timeline.  The code is not kept.
// A second attempt late in the round put the strips at the END of the grid, by the four wavefronts of a workgroup (trsm_strip_4w: 134
// registers, inside this kernel's two-workgroups-per-unit budget), behind counters fed by write-through first-column tiles (no release):
// bit-identical again, and every merged launch 15-19 us longer than the plain one where the separate panel-solve launch costs 9-11 --
// a round's uniform tasks end together, so the strips start when the launch would have been over: 2.610 ms against 2.523.  Not kept.)
__global__ __launch_bounds__(256, 2) void k_syrk_update2(double* __restrict__ S, double* __restrict__ y, int ld, int k, int nt,
                                                     double* __restrict__ Linv_next, double* __restrict__ ok, double* __restrict__ stall, int* __restrict__ flag, int unstaged,
                                                     int q_tiles)
{
    extern __shared__ double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0 = k + 1, mt = nt - j0;
    const int n_tiles = mt * (mt + 1) / 2;
    // q_tiles (round 4): the LAST q_tiles tiles are updated in quarters (64 x 64 per workgroup, 32 x 32 per wavefront) -- the host asks
    // for that when the half-tile tasks would end in a round that fills at most half of the 512 task slots, so that the round takes
    // half as long (what the whole-tile kernel does with n_q4).  Same sums in the same order: bit-identical.
    const int n_half_tiles = n_tiles - 1 - q_tiles;               // tile indices 1 .. n_half_tiles in halves
    const int n_task_wgs = 16 * ((n_half_tiles + 7) / 8);         // two workgroups (column halves) each, in groups of eight tiles
    const int first_q4 = NDIAG + n_task_wgs;
    const int first_rhs = first_q4 + 4 * q_tiles;
    const int bid = blockIdx.x;
    if (bid >= first_rhs) {
        const int i = k + 1 + (bid - first_rhs);
        if (i >= nt) return;
        const double* Lik = S + (size_t)(k * TILE) * ld + (size_t)i * TILE;
        const double* yk = y + (size_t)k * TILE;
        for (int r = tid; r < TILE; r += 256) {
            double acc = 0;
#pragma unroll 8
            for (int c = 0; c < TILE; ++c) acc = __builtin_fma(Lik[(size_t)c * ld + r], yk[c], acc);
            y[(size_t)i * TILE + r] -= acc;
        }
        return;
    }
    if (bid < NDIAG) {
        // 16x16 block u = 4 bid + wave of the lower triangle of the diagonal tile, (bi, bj), bi >= bj
        const int u = bid * 4 + wave;
        int bi, bj;
        tile_of_index(u, bi, bj);
        const int row0 = j0 * TILE + bi * NB, col0 = j0 * TILE + bj * NB;
        double4_t out[1][1];
        update_block<1, 32>(S, ld, k, row0, col0, lane, out);
        double* C = S + (size_t)(col0 + (lane >> 4)) * ld + row0 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) C[(size_t)(4 * r) * ld] = out[0][0][r];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (bid != 0) {
            // publish: all stores of the workgroup done -> agent-scope release -> count
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        // block 0: wait for the eight others (bounded spin; relaxed polls, one acquire), pull the tile, factor it
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NDIAG - 1 && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(1);
            if (spins >= (1 << 24)) *stall = 1.0;        // a producer never arrived: reported as a device error (never folded into "not positive definite")
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        double* A = sm;
        double* T = S + (size_t)(j0 * TILE) * ld + (size_t)j0 * TILE;
        load_tile_packed(A, T, ld, tid);
        __syncthreads();
        const bool failed = potrf_tile_lds<false, LayPacked>(A, sm + PACKED_TILE_DOUBLES, Linv_next, tid);
        store_tile_packed(T, A, ld, tid);
        if (tid == 0 && failed) *ok = 0.0;
        return;
    }
    if (bid >= first_q4) {
        const int qq = bid - first_q4;
        int rt, ct;
        tile_of_index(n_half_tiles + 1 + (qq >> 2), rt, ct);
        if (rt == ct && (qq & 3) == 2) return;                    // diagonal tile: rows 0-63 of columns 64-127 lie above the diagonal
        const int row0 = (j0 + rt) * TILE + (qq & 1) * 64 + (wave & 1) * 32;
        const int col0 = (j0 + ct) * TILE + ((qq >> 1) & 1) * 64 + (wave >> 1) * 32;
        double4_t out[2][2];
        update_block<2, 8>(S, ld, k, row0, col0, lane, out);
        double* C = S + (size_t)(col0 + (lane >> 4)) * ld + row0 + (lane & 15);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) C[(size_t)(a * 16 + 4 * r) * ld + b * 16] = out[a][b][r];
        return;
    }
    const int q0 = bid - NDIAG;
    // half of a tile: 128 rows x 64 columns, a wavefront 64 x 32, accumulators loaded from C.
    // Which half: workgroups go to the eight XCDs round-robin and every XCD has its own 4 MB L2, while the operands of a launch are
    // ONE tile column of L ((nt - k - 1) x 128 KB: 6 MB in the first columns).  Dealing tiles out in linear order makes every XCD
    // touch every row block of that column (L2 hit rate 56 %).  With bit 1 of `unstaged` set, XCD x (= workgroups with q0 % 8 == x)
    // instead takes the x-th CONTIGUOUS eighth of the half-tile tasks in band order (tile_of_band_order: bands of SYRK_BAND tile rows,
    // column by column inside a band): a band's row blocks stay in that L2 while its column blocks stream through once (hit rate 65 %,
    // memory-side reads -27 %) -- measured, and no faster (see chol_factor_solve), so linear order stays the default.  The two halves of
    // a tile are consecutive tasks of one XCD either way.  Placement only: any order gives the same numbers.
    int rt, ct, q;
    if (unstaged & 2) {
        const int n_task = 2 * n_half_tiles, per = (n_task + 7) / 8;
        q = (q0 & 7) * per + (q0 >> 3);
        if ((q0 >> 3) >= per || q >= n_task) return;
        tile_of_band_order(1 + (q >> 1), mt, rt, ct);
    } else {
        // (the two halves of a tile on one XCD, tiles in linear order: workgroups 16 g + t and 16 g + 8 + t serve tile 8 g + t)
        const int tile_i = 8 * (q0 >> 4) + (q0 & 7);
        if (tile_i >= n_half_tiles) return;
        q = 2 * tile_i + ((q0 >> 3) & 1);
        tile_of_index(1 + (q >> 1), rt, ct);
    }
    if (unstaged & 1) {                               // the default: operands straight from L2 per wavefront
        if (rt == ct && (q & 1) && (wave & 1) == 0) return;      // diagonal tile: rows 0-63 of columns 64-127 lie above the diagonal
        const int row0 = (j0 + rt) * TILE + (wave & 1) * 64, col0 = (j0 + ct) * TILE + (q & 1) * 64 + (wave >> 1) * 32;
        double4_t out[2][4];
        update_rect<2, 4, CHOL_RECT_KSTEPS, false, CHOL_RECT_NBUF>(S, ld, k, row0, col0, lane, out);
        double* C = S + (size_t)(col0 + (lane >> 4)) * ld + row0 + (lane & 15);
        if (unstaged & 8) {          // the updated block goes THROUGH to memory: nothing of it is left for the kernel boundary to write back
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) store_through(C + (size_t)(a * 16 + 4 * r) * ld + b * 16, out[a][b][r]);
            return;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) C[(size_t)(a * 16 + 4 * r) * ld + b * 16] = out[a][b][r];
        return;
    }
    if (unstaged & 4) update_half_tile_glds(S, ld, k, (j0 + rt) * TILE, (j0 + ct) * TILE + (q & 1) * 64, sm, tid);
    else update_half_tile_staged(S, ld, k, (j0 + rt) * TILE, (j0 + ct) * TILE + (q & 1) * 64, sm, tid);
}

// ---------------------------------------------------------------------------------------------
// backward substitution  L^T x = y  as ONE persistent launch: workgroup j owns tile column j.  It keeps L_jj (LDS) and
// y_j, consumes x_k for k = nt-1 .. j+1 as they appear, applying y_j -= L_kj^T x_k from a register-resident copy of
// the tile that was prefetched while it waited, then solves x_j = L_jj^-T y_j (blocked by 16 with the stored block
// inverses) and publishes it.  The tile-to-tile chain is pure latency, so there is no flag: x is pre-filled with a
// signalling-NaN pattern no computation produces, the producer stores its 128 values with agent-scope atomics and the
// consumers poll the VALUES (one L2 round trip per hop instead of flag + data: 6.1 -> 4.8 us per hop, measured with
// CHOL_DBG=1 tools/chol_test).  Inverting L_jj in LDS beforehand (one matrix-vector product at the end) was tried: the
// 50 us it takes do not fit in the slack of any column, and the hop stayed at 4.7 us; pinning the chain to one XCD did
// not shorten it either.  Progress relies on IN-ORDER WORKGROUP DISPATCH: a consumer (column j) is dispatched after every
// producer it waits for (columns > j come first in the grid), so a waiting workgroup never occupies a slot its producer
// needs -- with one handle all nt <= 256 workgroups are resident at once, with several handles on one GPU they may not be,
// and the dispatch order is then what guarantees progress.  Polls are bounded; a time-out sets *stall, which the host turns
// into MAGE_ERR_DEVICE (it is never folded into the "matrix not positive definite" outcome, which steers the LM loop).
// ---------------------------------------------------------------------------------------------


__global__ __launch_bounds__(256) void k_bsolve_persist(const double* __restrict__ S, const double* __restrict__ y, double* __restrict__ x,
                                                        int ld, int nt, const double* __restrict__ Linv, double* __restrict__ stall, long long* __restrict__ dbg)
{
    extern __shared__ double sm[];
    constexpr int LDB = TILE + 2;
    double* T = sm;                        // first L_jj^-T, then the off-diagonal tile staged for the NEXT arrival; column-major, pitch LDB
    __shared__ double ys[TILE], xk[TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = nt - 1 - (int)blockIdx.x;                     // the last tile column is dispatched first
    if (tid < TILE) ys[tid] = y[(size_t)j * TILE + tid];
    // The tile's inverse by the panel solve's strip code on the rows of the identity (two strips per wavefront, ~10 us, while every
    // column but the last two is waiting anyway): the end of a hop is then one product from registers, not eight substitution steps.
    trsm_strip<true>(const_cast<double*>(S), nullptr, ld, j, wave, false, Linv + (size_t)j * NBLK * NB * NB, lane, T, LDB);
    trsm_strip<true>(const_cast<double*>(S), nullptr, ld, j, wave + 4, false, Linv + (size_t)j * NBLK * NB * NB, lane, T, LDB);
    __syncthreads();
    // thread (c, half) owns rows half * 64 .. +63 of column c of the current off-diagonal tile, and columns half * 64 .. of row c of the inverse
    const int c = tid >> 1, half = tid & 1;
    double minv[64], tile[64];
#pragma unroll
    for (int q = 0; q < 64; ++q) minv[q] = T[(half * 64 + q) * LDB + c];
    __syncthreads();                       // T is free
    // The solve is a pipeline clocked by how fast a column consumes the x_k, one 128-KB tile each, and a wavefront's loads return in
    // order (a poll issued behind a tile fetch cannot return before it has landed).  So the tiles travel global -> LDS by the
    // load-to-LDS path of wavefronts 2-3 (one fully coalesced 1-KB column per instruction, no registers), a whole arrival ahead;
    // every thread takes its share from LDS into registers; and wavefronts 0-1, which poll, never have a tile load in flight.
    auto stage = [&](int k) {              // wavefronts 2-3: tile (k, j) -> T
        const double* g = S + (size_t)(j * TILE + (wave - 2) * 64) * ld + (size_t)k * TILE + lane * 2;
#pragma unroll 8
        for (int q = 0; q < 64; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)q * ld),
                                             (__attribute__((address_space(3))) void*)(T + ((wave - 2) * 64 + q) * LDB), 16, 0, 0);
    };
    auto take = [&]() {                    // T -> registers
#pragma unroll
        for (int q = 0; q < 64; ++q) tile[q] = T[c * LDB + half * 64 + q];
    };
    if (j < nt - 1) {
        if (wave >= 2) { stage(nt - 1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __syncthreads();
        take();
        __syncthreads();
        if (wave >= 2 && nt - 2 > j) stage(nt - 2);
    }
    if (dbg && tid == 0) { dbg[j * 4 + 0] = wall_clock64(); dbg[j * 4 + 1] = dbg[j * 4 + 0]; }
    for (int k = nt - 1; k > j; --k) {
        if (k == j + 1 && dbg && tid == 0) dbg[j * 4 + 2] = wall_clock64();
        if (tid < TILE) {
            const unsigned long long* src = reinterpret_cast<const unsigned long long*>(x + (size_t)k * TILE + tid);
            unsigned long long v;
            int spins = 0;
            while ((v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == X_SENTINEL && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
            if (v == X_SENTINEL) *stall = 3.0;           // the producer column never published: reported as a device error
            xk[tid] = __longlong_as_double((long long)v);
        }
        __syncthreads();
        double a0 = 0, a1 = 0;
#pragma unroll
        for (int q = 0; q < 64; q += 2) {
            a0 = __builtin_fma(tile[q], xk[half * 64 + q], a0);
            a1 = __builtin_fma(tile[q + 1], xk[half * 64 + q + 1], a1);
        }
        double acc = a0 + a1;
        acc += __shfl_xor(acc, 1, 64);
        if (half == 0) ys[c] -= acc;
        if (k - 1 > j) {
            if (wave >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // tile k - 1 has landed in T
            __syncthreads();
            take();
            __syncthreads();
            if (wave >= 2 && k - 2 > j) stage(k - 2);
        } else __syncthreads();
    }
    {   // x_j = L_jj^-T ys
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
        for (int q = 0; q < 64; q += 4) {
            a0 = __builtin_fma(minv[q + 0], ys[half * 64 + q + 0], a0); a1 = __builtin_fma(minv[q + 1], ys[half * 64 + q + 1], a1);
            a2 = __builtin_fma(minv[q + 2], ys[half * 64 + q + 2], a2); a3 = __builtin_fma(minv[q + 3], ys[half * 64 + q + 3], a3);
        }
        double acc = (a0 + a1) + (a2 + a3);
        acc += __shfl_xor(acc, 1, 64);
        if (half == 0)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(x + (size_t)j * TILE + c), (unsigned long long)__double_as_longlong(acc),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (dbg && tid == 0) dbg[j * 4 + 3] = wall_clock64();
}

// ---------------------------------------------------------------------------------------------
// Small systems (order n <= 128: local bundle adjustment, map initialisation, pose-only refinement): the whole dense solve in
// ONE launch of one workgroup -- the leading ceil(n / 16) blocks of the tile factored in LDS (potrf_tile_lds<true>), then both
// substitutions in LDS with the 16x16 block inverses.  The chain of launches of the large-system path (potrf + solve + backward
// solve, ~40 us for any n <= 128) becomes ~3 us per 16 columns.
// ---------------------------------------------------------------------------------------------
// (INV_IN_LDS is a template parameter, not an argument: chosen at run time, `Linv` would be a generic pointer and every read of a
// block inverse in the substitutions a flat load)
template <bool INV_IN_LDS>
__global__ __launch_bounds__(256) void k_small_solve(const double* __restrict__ S, const double* __restrict__ y, double* __restrict__ x, int n, int ld,
                                                     double* __restrict__ Linv_ws, double* __restrict__ ok, double* __restrict__ stall)
{
    extern __shared__ double sm[];
    const int tid = threadIdx.x;
    const int nblk = (n + NB - 1) / NB, np = nblk * NB;          // S carries the identity beyond n (it is padded to a whole tile)
    // LDS: the leading np columns of the tile, the two block inverses the factorisation juggles and -- when it fits (np <= 112) --
    // all block inverses, so that the substitutions never leave LDS (from the global workspace every block step paid an L2 round trip)
    double* A = sm;
    double* Li = sm + np * LDC;
    double* const Linv = INV_IN_LDS ? Li + 2 * NB * NB : Linv_ws;
    __shared__ double rhs[TILE], xc[NB];
    {   // column-major copy, two doubles per load (np is a multiple of 16), eight loads in flight per thread
        const int half = np / 2, total = np * half;
        for (int e0 = tid; e0 < total; e0 += 256 * 8) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {          // (every element assigned on every path: a conditionally initialised array went to scratch memory)
                const int e = e0 + 256 * u, ec = e < total ? e : 0, c = ec / half, r2 = ec - c * half;
                v[u] = *reinterpret_cast<const double2*>(S + (size_t)c * ld + 2 * r2);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int e = e0 + 256 * u; if (e < total) { const int c = e / half, r2 = e - c * half; *reinterpret_cast<double2*>(A + c * LDC + 2 * r2) = v[u]; } }
        }
    }
    for (int i = tid; i < np; i += 256) rhs[i] = y[i];
    __syncthreads();
    const bool failed = potrf_tile_lds<true, LayLDC>(A, Li, Linv, tid, nblk);
    __threadfence_block();
    __syncthreads();
    // forward substitution L z = rhs (four partial sums per product: a dependent f64 FMA costs ~25 cycles on a lone wavefront)
    for (int cb = 0; cb < nblk; ++cb) {
        if (tid < NB) {
            double p[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int q = 0; q < NB; ++q) p[q & 3] = __builtin_fma(Linv[cb * NB * NB + tid * NB + q], rhs[cb * NB + q], p[q & 3]);
            xc[tid] = (p[0] + p[1]) + (p[2] + p[3]);
        }
        __syncthreads();
        if (tid < NB) rhs[cb * NB + tid] = xc[tid];
        const int r = (cb + 1) * NB + tid;
        if (r < np) {
            double p[4] = { rhs[r], 0, 0, 0 };
#pragma unroll
            for (int q = 0; q < NB; ++q) p[q & 3] = __builtin_fma(-A[(cb * NB + q) * LDC + r], xc[q], p[q & 3]);
            rhs[r] = (p[0] + p[1]) + (p[2] + p[3]);
        }
        __syncthreads();
    }
    // backward substitution L^T x = z
    for (int cb = nblk - 1; cb >= 0; --cb) {
        if (tid < NB) {
            double p[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int q = 0; q < NB; ++q) p[q & 3] = __builtin_fma(Linv[cb * NB * NB + q * NB + tid], rhs[cb * NB + q], p[q & 3]);
            xc[tid] = (p[0] + p[1]) + (p[2] + p[3]);
        }
        __syncthreads();
        if (tid < NB) rhs[cb * NB + tid] = xc[tid];
        if (tid < cb * NB) {
            double p[4] = { rhs[tid], 0, 0, 0 };
#pragma unroll
            for (int q = 0; q < NB; ++q) p[q & 3] = __builtin_fma(-A[tid * LDC + cb * NB + q], xc[q], p[q & 3]);
            rhs[tid] = (p[0] + p[1]) + (p[2] + p[3]);
        }
        __syncthreads();
    }
    for (int i = tid; i < n; i += 256) x[i] = rhs[i];
    if (tid == 0) { *ok = failed ? 0.0 : 1.0; *stall = 0.0; }
}


}  // namespace

// per tile column: the inverses of its eight diagonal blocks, then (behind all of those) the scratch copy of its 28 sub-diagonal blocks
// (the third part: per tile column the full inverse of its diagonal tile, 36 blocks -- k_syrk_update<1, true, true>)
size_t chol_workspace_doubles(int n_pad) { return (size_t)(n_pad / TILE) * (NBLK * NB * NB + LPUB_TILE_DOUBLES + PACKED_TILE_DOUBLES); }

// Kernels that need more than the default dynamic-LDS limit must be opted in once per device (function attributes
// are per device); called from mage_ba_create after hipSetDevice.
int g_n_cu = 256;        // compute units of the device the library was initialised on (gfx950: 256)

void chol_debug_syrk_stamps(long long* out32, bool reset)
{
    if (reset) {
        long long init[32];
        for (int i = 0; i < 32; ++i) init[i] = 0;
        init[6] = 0x7fffffffffffffffLL;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_syrk_dbg), init, sizeof(init));
    } else (void)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_syrk_dbg), 32 * sizeof(long long));
}

bool chol_merge_fallback_active() { return g_merge_disabled.load(std::memory_order_relaxed); }

void chol_report_stall(int code)
{
    if (code == 2) g_merge_disabled.store(true, std::memory_order_relaxed);
}

void chol_init_device()
{
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        g_n_cu = prop.multiProcessorCount;
    const size_t lds_diag = ((size_t)PACKED_TILE_DOUBLES + 2 * NB * NB) * sizeof(double);
    const size_t lds_small = ((size_t)TILE * LDC + 2 * NB * NB) * sizeof(double);
    const size_t lds_panel = (size_t)TILE * (TILE + 2) * sizeof(double);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_diag), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_diag);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_update<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_diag);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_update<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_diag);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_update<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_diag);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_update<1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_diag + PACKED_TILE_DOUBLES * sizeof(double)));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_update<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_diag);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_update2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_diag);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bsolve_persist), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_panel);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_solve<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_solve<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small);
}

void chol_small_solve(const double* S, const double* y, double* x, int n, int ld, double* Linv_ws, double* ok, double* stall, hipStream_t st)
{
    const int nblk = (n + NB - 1) / NB, np = nblk * NB;
    const size_t with_inv = ((size_t)np * LDC + 2 * NB * NB + (size_t)nblk * NB * NB) * sizeof(double);
    const bool inv_in_lds = with_inv + 4096 <= 160 * 1024;          // + the kernel's static LDS
    const size_t lds = inv_in_lds ? with_inv : ((size_t)np * LDC + 2 * NB * NB) * sizeof(double);
    if (inv_in_lds) hipLaunchKernelGGL(k_small_solve<true>, dim3(1), dim3(256), lds, st, S, y, x, n, ld, Linv_ws, ok, stall);
    else hipLaunchKernelGGL(k_small_solve<false>, dim3(1), dim3(256), lds, st, S, y, x, n, ld, Linv_ws, ok, stall);
}

// Right-looking factorisation.  Step k = one k_trsm_panel launch + one k_syrk_update launch; the diagonal
// factorisation of step k+1 is done by workgroup 0 of step k's update (tile (k+1, k+1) is the first tile of
// the grid), which hides most of the sequential diagonal work behind the bulk of the update.
// (Round 4 also tried LOOK-AHEAD ACROSS STREAMS in the update-bound columns: the update of column k as two launches that depend only on
// panel k and touch disjoint tiles -- tile column k + 1 with its diagonal tile's factorisation and then its panel solve on a side stream of
// the highest priority, the rest of the trailing matrix on the caller's stream, events between them -- so that the ~10.6 us panel-solve
// launches would run beside the large updates.  Bit-identical, and 2.93-2.96 ms against 2.485: the large update fills every compute unit
// with two 78 KB / 239-register workgroups and the dispatcher serves the older launch first, so the column launch (25 us alone) got its
// slots only as the large update drained -- 118 us beside a 102 us update -- and the chain was serial again, plus the events.  Not kept;
// profiles/HISTORY.md has the per-launch timeline.)
void chol_factor_solve(double* S, double* y, double* x, int n_pad, const CholWorkspace& ws, double* ok, hipStream_t st)
{
    const int nt = n_pad / TILE;
    const size_t lds_diag = ((size_t)PACKED_TILE_DOUBLES + 2 * NB * NB) * sizeof(double);
    const size_t lds_panel = (size_t)TILE * (TILE + 2) * sizeof(double);
    const size_t linv_stride = (size_t)NBLK * NB * NB;
    double* stall = ws.stall ? ws.stall : ok + 1;     // callers without a slot of their own pass a two-element ok
    hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(256), lds_diag, st, S, n_pad, 0, ws.Linv, ok, stall, reinterpret_cast<unsigned long long*>(x), n_pad);
    // Column 0's panel solve is a launch of its own (it also zeroes the hand-off counters).  After that, step k = the trailing update
    // by column k, which factors the diagonal tile of column k + 1 inside -- and, in the whole-tile form, also SOLVES column k + 1's
    // panel inside (strips at the end of the grid, waiting for the factored tile and for the first-column tiles of this very launch):
    // while the update dominates the strips run beside its last tiles, afterwards they save the launch boundary (~10 us per column
    // on the chain either way).  The half-tile form is followed by a panel-solve launch as before.
    // (g_merge_disabled: a strip's wait ran out once in this process -- chol_report_stall -- which only happens when SEVERAL PROCESSES
    // share the GPU: the hardware scheduler then saves and restores workgroups, and the waiting strips, one per compute unit because of
    // the launch's 151 KB of LDS, can keep the producers they wait for from being restored.  From then on the panel solve is its own launch.)
    static const bool merge_env_off = std::getenv("MAGE_CHOL_NO_MERGED_TRSM") != nullptr;
    const bool merge_off = merge_env_off || g_merge_disabled.load(std::memory_order_relaxed);
    static const bool pipelined_fill = std::getenv("MAGE_CHOL_PIPELINED_TRSM") != nullptr;
    // Strips by the four wavefronts of their workgroup (trsm_strip_4w): OFF by default, MAGE_CHOL_STRIP_4W=1 selects it.  Measured (round 4,
    // tools/_bin/chol_test 6016, time stamps of launch 36): the strip ends 6.4 us after L_kk is stored instead of 7.2 -- its 4.7 us of
    // matrix-core issue were never the long leg, the flag, the operands' trip past the L2 and the store are -- and the panel-solve launches
    // gain nothing (10.8 against 10.6 us: 256-thread workgroups, seven barriers): 2.577-2.583 ms against 2.566-2.572.  Bit-identical.
    static const bool strip4w = std::getenv("MAGE_CHOL_STRIP_4W") != nullptr;
    if (strip4w) hipLaunchKernelGGL(k_trsm_panel, dim3((nt - 1) * NBLK + 1), dim3(256), 0, st, S, y, n_pad, 0, nt, ws.Linv, ws.sync, pipelined_fill ? -1 : 0);
    else hipLaunchKernelGGL(k_trsm_panel_1w, dim3((nt - 1) * NBLK + 1), dim3(64), 0, st, S, y, n_pad, 0, nt, ws.Linv, ws.sync, pipelined_fill ? -1 : 0);
    int col_total = 0;
    static const int dbg_col = std::getenv("CHOL_DBG_COL") ? std::atoi(std::getenv("CHOL_DBG_COL")) : -1;
    for (int k = 0; k + 1 < nt; ++k) {
        const int m = nt - k - 1;             // tile rows below panel k = tile rows of the trailing matrix
        const int n_tiles = m * (m + 1) / 2;
        // Quarter tiles in the one-workgroup-per-unit kernel: a last round that fills at most half of the compute units (syrk_quartered_tiles),
        // and -- late in round 4 -- EVERY tile of the chain-bound columns: a whole tile is one wavefront per SIMD streaming 128 KB of
        // operands with a single chunk of prefetch, and with 130-250 of them in flight it takes 24-44 us; in quarters the same update ends
        // before the chain does (2.520-2.527 -> 2.496-2.501 ms per factorisation, bit-identical).  MAGE_CHOL_TAIL_WHOLE_TILES=1: the old rule.
        static const bool tail_whole = std::getenv("MAGE_CHOL_TAIL_WHOLE_TILES") != nullptr;
        int n_q4 = syrk_quartered_tiles(n_tiles, g_n_cu);
        // Two forms of the trailing update.  While it is what takes the time (more than ~1.5 rounds of whole tiles) the half-tile
        // form runs two workgroups per compute unit (239 registers, the packed 78 KB of LDS): 94 / 86 / 84 us on the first columns
        // against 104 / 91 / 89.  Once the chain (diagonal update, hand-off, in-tile factorisation: ~29 us) is what takes the time,
        // the workgroup that factors must not share its compute unit: the whole-tile form (382 registers: one workgroup per unit).
        static const int bulk2_min_tiles = std::getenv("MAGE_CHOL_BULK2_MIN_TILES") ? std::atoi(std::getenv("MAGE_CHOL_BULK2_MIN_TILES")) : 400;
        const bool bulk2 = n_tiles >= bulk2_min_tiles;
        if (!bulk2 && !tail_whole) n_q4 = n_tiles - 1;
        // the half-tile task in three forms: operands per wavefront straight from L2 (default), staged through LDS by plain loads + ds_write
        // (MAGE_CHOL_BULK2_STAGED=1: measured slower in round 2, 3.01 ms against 2.87), staged by the load-to-LDS path behind raw barriers
        // (MAGE_CHOL_BULK2_FORM=glds, round 4)
        static const bool form_glds = std::getenv("MAGE_CHOL_BULK2_FORM") && std::string(std::getenv("MAGE_CHOL_BULK2_FORM")) == "glds";
        static const bool unstaged = std::getenv("MAGE_CHOL_BULK2_STAGED") == nullptr && !form_glds;
        bool merged = false;
        // Band placement (tile_of_band_order) is OFF by default: it does what it is for -- L2 hit rate of the launch 56 -> 65 %, memory-side
        // reads -27 % (profiles/r03_chol_pmc.txt) -- but the matrix cores are busy 49.7 % of the launch either way and the factorisation
        // takes 2.731 ms against 2.710: the update is not waiting for its operands' misses.  MAGE_CHOL_XCD_BANDS=1 selects it.
        static const bool xcd_bands = std::getenv("MAGE_CHOL_XCD_BANDS") != nullptr;
        // Updated tiles (half and whole tiles) are stored THROUGH: they are read again only after the kernel boundary, and what was written
        // through is not left for the boundary to write back (2.536-2.548 -> 2.517-2.528 ms per factorisation, three A/B pairs; bit-identical).
        // MAGE_CHOL_WT_TILES=0: plain stores.
        static const bool wt_tiles = !(std::getenv("MAGE_CHOL_WT_TILES") && std::atoi(std::getenv("MAGE_CHOL_WT_TILES")) == 0);
        if (bulk2) {
            // tiles of the last, partial round of half-tile tasks go in quarters when that round fills at most half of the task slots
            // (two workgroups on each compute unit); MAGE_CHOL_NO_QUARTERS=1 switches it off
            static const bool quarters_off = std::getenv("MAGE_CHOL_NO_QUARTERS") != nullptr;
            const int tiles_per_round = g_n_cu;                       // 2 g_n_cu task slots, two half-tile tasks per tile
            const int rem_tiles = (n_tiles - 1) % tiles_per_round;
            const int q_tiles = (!quarters_off && unstaged && !xcd_bands && rem_tiles > 0 && rem_tiles * 2 <= tiles_per_round) ? rem_tiles : 0;
            hipLaunchKernelGGL(k_syrk_update2, dim3(NDIAG + 16 * ((n_tiles - 1 - q_tiles + 7) / 8) + 4 * q_tiles + m), dim3(256), lds_diag, st, S, y, n_pad, k, nt,
                               ws.Linv + (size_t)(k + 1) * linv_stride, ok, stall, ws.sync, (unstaged ? 1 : 0) | (xcd_bands ? 2 : 0) | (form_glds ? 4 : 0) | (wt_tiles ? 8 : 0), q_tiles);
        }
        else {
            merged = !merge_off;
            const int n_whole = n_tiles - 1 - n_q4;
            if (merged) for (int rt = 1; rt < m; ++rt) col_total += (rt * (rt + 1) / 2 <= n_whole) ? 1 : 4;      // workgroups that write a part of column k + 1
            // Pipelined strips (trsm_strip_pipelined) are OFF by default: measured (CHOL_DBG_COL time stamps, tools/chol_test) they end
            // 3.5 us after the tile is factored instead of 9.9 in the last columns (launch 38.8 -> 36.8 us), but every strip then reads
            // the same 56 KB of L_kk past the L2 (agent-scope loads; 200 strips in the first chain-bound columns: launch 41.8 -> 52.8 us)
            // and their polling takes fabric bandwidth from the factoring workgroup (tile load 1.4 -> 3.6 us): 2.80 ms per factorisation
            // against 2.71.  One release + one acquire per strip and L2-cached reads of the broadcast operand stay.
            // (MAGE_CHOL_PIPELINED_TRSM=<rows>: only in the columns with at most <rows> tile rows left, where few strips poll; 1 = every column)
            static const int pipelined_rows = std::getenv("MAGE_CHOL_PIPELINED_TRSM") ? std::atoi(std::getenv("MAGE_CHOL_PIPELINED_TRSM")) : 0;
            const bool pipelined = pipelined_rows == 1 || (pipelined_rows > 1 && m <= pipelined_rows);
            // the write-through form of the chain's two hand-offs (k_syrk_update<1, true>); MAGE_CHOL_WT_HANDOFF=0 restores release / acquire
            static const bool wt_handoff = !(std::getenv("MAGE_CHOL_WT_HANDOFF") && std::atoi(std::getenv("MAGE_CHOL_WT_HANDOFF")) == 0);
            // Phased strips (trsm_strip_phased, round 4): ON by default -- three polls of ONE progress word behind the in-tile factorisation instead of
            // a wait for the whole tile.  Measured (tools/_bin/chol_test 6016 10, time stamps of launches 24 / 36 / 44): the last strip ends 3.8 us
            // after the tile is factored instead of 8.5 (the in-tile factorisation itself 20.8 against 20.2 us), 2.527-2.530 ms per factorisation
            // against 2.570-2.577; bit-identical.  MAGE_CHOL_PHASED_TRSM=0 switches it off, =<rows> restricts it to the columns with at most
            // <rows> tile rows left (no better: 16 / 20 / 24 rows 2.535 / 2.531 / 2.527).
            static const int phased_rows = std::getenv("MAGE_CHOL_PHASED_TRSM") ? std::atoi(std::getenv("MAGE_CHOL_PHASED_TRSM")) : 1;
            const bool phased = wt_handoff && !pipelined && (phased_rows == 1 || (phased_rows > 1 && m <= phased_rows));
            // The strips as products over the tile's full inverse (k_syrk_update<1, true, true>): OFF by default, MAGE_CHOL_GEMM_STRIPS=1 selects
            // it.  Measured (time stamps of launch 30, tools/_bin/chol_test 6016, profiles/r04_chol_links.txt): the strips do what they were
            // built for -- last strip done 3.4 us after the flag instead of 7.1 -- but building the inverse beside the in-tile factorisation
            // stretches THAT from 20.2 to 30.7 us (the three helper wavefronts no longer finish inside wavefront 0's 2 us per pivot block;
            // not the instruction cache: SQC_ICACHE_MISSES 1.1 k per launch either way), so a chain-bound column takes 40.1 us against
            // 35.8 and the factorisation 2.73 ms against 2.60.  Same residual (7.5e-16), all tests pass; kept for the record.
            static const bool gemm_strips = std::getenv("MAGE_CHOL_GEMM_STRIPS") && std::atoi(std::getenv("MAGE_CHOL_GEMM_STRIPS")) != 0;
            const dim3 grid(NDIAG + n_whole + 4 * n_q4 + m + (merged ? (m - 1) * NBLK : 0));
            const int dbg = (ws.dbg && dbg_col == k) ? 1 : 0;
            double* const Linv_next = ws.Linv + (size_t)(k + 1) * linv_stride;
            double* const Lpub_next = ws.Linv + (size_t)nt * linv_stride + (size_t)(k + 1) * LPUB_TILE_DOUBLES;
            if (!merged) hipLaunchKernelGGL(k_syrk_update<0>, grid, dim3(256), lds_diag, st, S, y, n_pad, k, nt, Linv_next, ok, stall, ws.sync, n_q4, col_total, dbg, Lpub_next, strip4w ? 1 : 0);
            else if (phased && !gemm_strips) hipLaunchKernelGGL((k_syrk_update<3, true>), grid, dim3(256), lds_diag, st, S, y, n_pad, k, nt, Linv_next, ok, stall, ws.sync, n_q4, col_total, dbg, Lpub_next, wt_tiles ? 2 : 0);
            else if (pipelined) hipLaunchKernelGGL(k_syrk_update<2>, grid, dim3(256), lds_diag, st, S, y, n_pad, k, nt, Linv_next, ok, stall, ws.sync, n_q4, col_total, dbg, Lpub_next, strip4w ? 1 : 0);
            else if (wt_handoff && gemm_strips)
                hipLaunchKernelGGL((k_syrk_update<1, true, true>), grid, dim3(256), lds_diag + PACKED_TILE_DOUBLES * sizeof(double), st, S, y, n_pad, k, nt, Linv_next, ok, stall, ws.sync,
                                   n_q4, col_total, dbg, ws.Linv + (size_t)nt * (linv_stride + LPUB_TILE_DOUBLES) + (size_t)(k + 1) * PACKED_TILE_DOUBLES, strip4w ? 1 : 0);
            else if (wt_handoff) hipLaunchKernelGGL((k_syrk_update<1, true>), grid, dim3(256), lds_diag, st, S, y, n_pad, k, nt, Linv_next, ok, stall, ws.sync, n_q4, col_total, dbg, Lpub_next, strip4w ? 1 : 0);
            else hipLaunchKernelGGL(k_syrk_update<1>, grid, dim3(256), lds_diag, st, S, y, n_pad, k, nt, Linv_next, ok, stall, ws.sync, n_q4, col_total, dbg, Lpub_next, strip4w ? 1 : 0);
        }
        if (!merged && strip4w) hipLaunchKernelGGL(k_trsm_panel, dim3((m - 1) * NBLK + 1), dim3(256), 0, st, S, y, n_pad, k + 1, nt, ws.Linv + (size_t)(k + 1) * linv_stride, ws.sync, 0);
        else if (!merged) hipLaunchKernelGGL(k_trsm_panel_1w, dim3((m - 1) * NBLK + 1), dim3(64), 0, st, S, y, n_pad, k + 1, nt, ws.Linv + (size_t)(k + 1) * linv_stride, ws.sync, 0);
    }
    // backward substitution: one persistent launch (needs every workgroup resident: nt <= 256 compute units)
    hipLaunchKernelGGL(k_bsolve_persist, dim3(nt), dim3(256), lds_panel, st, S, y, x, n_pad, nt, ws.Linv, stall, ws.dbg);
}

}  // namespace mage
