// chol_device.h -- device-side building blocks of the dense float64 Cholesky on gfx950, shared by the column-by-column
// launches (chol_kernels.hip) and the one-launch task-graph schedule (chol_dag.hip).  chol_kernels.hip describes the algorithm,
// the operand / accumulator layout of v_mfma_f64_16x16x4_f64 and why every triangular solve is written on the transposed unknown.
#pragma once
#include <hip/hip_runtime.h>
#include "chol_kernels.h"

namespace mage {
namespace chol {

constexpr int TILE = CHOL_TILE;      // 128
constexpr int NB = 16;               // inner block (one MFMA tile)
constexpr int NBLK = TILE / NB;      // 8
constexpr int LDC = TILE + 16;       // LDS column pitch (doubles): consecutive k columns land 32 banks apart
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr unsigned long long X_SENTINEL = 0x7FF4DEADBEEF0001ull;   // "not published yet" in x (k_bsolve_persist): a signalling-NaN pattern no computation produces


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
// 16x16 diagonal block: Cholesky in registers by one wavefront, together with a second register set x[] that obeys the same recurrence.
// Lane l (mod 16; the four 16-lane rows of the wavefront hold identical copies of a[]) owns ROW l of the block (a[c] = A[l][c]); x[] is,
// per 16-lane row, COLUMN l of L^-1 (x[c] = Linv[c][l], from a row of the identity) or ROW l of a block below the pivot block (which
// comes out as the strip solved by substitution): factor_block16_rows, below.  Both recurrences are
//     a[c] -= a[j] * L[c][j],   x[c] -= x[j] * L[c][j]      (c > j),        a[j], x[j] *= 1 / L[j][j]
// and the coefficient L[c][j] is lane c's a[j]: it enters the FMA as a DPP operand (row_newbcast:c -- the only DPP
// control gfx90a+ allows on 64-bit operations, and exactly the one needed), so an update is ONE v_fmac_f64_dpp with no
// trip through the SGPR file.  Inline asm is invisible to the hazard recogniser: "VALU writes VGPR -> DPP reads it" needs
// 2 wait states, supplied by explicit s_nop where a DPP source was written just before.
// A non-positive pivot leaves NaNs behind and is reported once, at the end; the caller discards the factorisation.
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

// tools/f64_latency.hip: on a lone wavefront a dependent v_fma_f64 returns after 8 cycles, v_rsq_f64 after 20, a DPP operand after 16
// -- and ANY f64 operation, DPP or not, takes 4.8 cycles of issue: the block is bound by instruction COUNT.  A pivot is the plain
// recurrence -- broadcast the diagonal entry, 1 / sqrt(d), scale the column, update the columns behind it -- 13 + 2 (15 - J)
// instructions, ~450 per block; the previous pivot's updates are dealt into the latency slots of this pivot's chain.  (The
// round-3 form that shortened the recurrence instead -- two pivots ahead, ~620 instructions -- is in git history: 3 950 against
// 3 520 cycles per block.)
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

// ---------------------------------------------------------------------------------------------
// diagonal tile, LDS-resident, blocked by 16 (potrf_tile_rows, below).
// PUBLISH: 0 plain stores of the block inverses (a kernel boundary follows); 1 the block inverses written THROUGH (agent-scope
// stores, no release fence follows); 4 = 1 plus the PHASED hand-off: the sub-diagonal blocks of every finished block column go out to a
// scratch copy of the tile (LPUB_TILE_DOUBLES per tile, block (c, j) at (c (c - 1) / 2 + j) * 256, column-major) with stores that go
// through to memory, and ONE progress word is raised an in-tile iteration later -- by then the stores have long been acknowledged, so
// the publisher never waits (a flag right behind the stores cost +2 ... +6 us per tile on the critical path).  The strips poll that
// word (trsm_strip_phased, strips_phased of chol_dag.hip).
// (Forms built and measured slower in rounds 3-6 -- strips polling the operands for a sentinel; the tile's full inverse built alongside;
// the round-5 "lean" form with the strips as products with the block inverse BEHIND the pivot recurrence, 42 970 cycles per tile against
// this form's 34 950; every row of the tile in one v_readlane recurrence -- profiles/HISTORY.md has the numbers, git the code.)
constexpr int LPUB_BLOCKS = NBLK * (NBLK - 1) / 2;            // 28 sub-diagonal blocks
constexpr int LPUB_TILE_DOUBLES = LPUB_BLOCKS * NB * NB;      // 7168
struct TilePublish {
    double* Lpub = nullptr;       // this tile's scratch blocks
    int* progress = nullptr;      // one word: base + c once the blocks of columns <= c (and the block inverses <= c + 1) are in memory
    int base = 0;                 // (8 x tile index: the word only ever grows inside a factorisation)
};

// The barriers inside the tile factorisation order LDS traffic only.  __syncthreads() also waits for the wavefront's outstanding
// GLOBAL stores (the compiler puts s_waitcnt vmcnt(0) in front of the barrier), and in the publishing forms those are write-through
// stores that take ~2 us to be acknowledged.  LDS_ONLY: wait for this wavefront's LDS operations, then the hardware barrier; global
// stores stay in flight (the caller drains them once, before it raises the flag).
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
// the matching load: not served from this compute unit's L1 (MI355X_MICROARCH.md, inter-workgroup visibility)
__device__ __forceinline__ double load_through(const double* p)
{
    return __hip_atomic_load((const global_double*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef CHOL_TILE_STAMPS        // tools/potrf_probe.hip: shader-clock stamps of every wavefront at the five points of an in-tile iteration
__device__ long long g_tile_stamps[8][NBLK][5];
#define TILE_STAMP(p) do { if (lane == 0) g_tile_stamps[wave][s][p] = clock64(); } while (0)
#else
#define TILE_STAMP(p) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// The tile factorisation with the rows BELOW the pivot block carried through the pivot recurrence (round 6).
// fl_column's second register set x[] obeys  x[c] -= x[j] L[c][j],  x[j] *= 1 / L[j][j]  with the coefficient taken from the lane's own
// 16-lane row (row_newbcast), and the four rows of a wavefront hold identical copies of a[].  So x[] may hold something DIFFERENT in
// each of the four rows at no cost in instructions: a row of the identity gives a column of the block inverse (what round 5's lean form
// computes four times over), and row l of a sub-diagonal block B(i, s) gives row l of Y(i, s) = B(i, s) L_ss^-T -- the strip, solved by
// substitution inside the recurrence instead of as a product with the inverse behind it.  Two wavefronts carry all of an in-tile step:
//     wavefront 0   rows: strip 0 | identity | strip 1 | strip 2          wavefront 1   rows: strips 3 | 4 | 5 | 6
// (wavefront 1 repeats the pivot block's own recurrence: redundant, but beside wavefront 0, not behind it).  What the lean form had on
// the critical path behind the pivot block -- barrier, four dependent matrix operations for strip 0, four more for the next diagonal
// block, an LDS round trip either side -- shrinks to: barrier, ONE set of four matrix operations per block of column s + 1 (the k = s
// term; the terms k < s were summed into the block by the idle wavefronts during the recurrence, left-looking), barrier.
// The strips' bits differ from the lean form's (a substitution instead of a product with a rounded inverse: the residual is that of a
// textbook trsm); the block inverses are the lean form's to the bit.
template <int PITCH>
__device__ __forceinline__ bool factor_block16_rows(double* __restrict__ A, int boff, int poff, int pst, int lane, bool store_l, double (&a)[NB])
{
    const int l = lane & 15;
    double* __restrict__ B = A + boff;
    double* __restrict__ P = A + poff;          // this lane's payload row: element c at P[c * pst]
    double x[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) { a[c] = B[c * PITCH + l]; x[c] = P[c * pst]; }
    double dmin = 1.0;
    fl_columns<0>(a, x, dmin);
    dmin = (l == NB - 1 && !(fabs(a[NB - 1]) < __builtin_huge_val())) ? -1.0 : 1.0;
    // (store_l false: another wavefront repeats this recurrence and may not have read the block yet -- the caller stores a[] behind a barrier)
    if (store_l && lane < NB) {
#pragma unroll
        for (int c = 0; c < NB; ++c) B[c * PITCH + l] = a[c];
    }
#pragma unroll
    for (int c = 0; c < NB; ++c) P[c * pst] = x[c];
    return __builtin_amdgcn_ballot_w64(!(dmin > 0.0)) != 0;
}

// two blocks of column bj in one go: (bi0, bj) and (bi1, bj) -= X(bi, bp) X(bj, bp)^T, the operand of bj shared, the two chains interleaved
template <class LAY>
__device__ __forceinline__ void lds_update_tile2(double* __restrict__ A, int bi0, int bi1, int bj, int bp, int lane)
{
    double* C0 = A + LAY::blk(bi0, bj);
    double* C1 = A + LAY::blk(bi1, bj);
    const double* Xa = A + LAY::blk(bj, bp);
    const double* Xb0 = A + LAY::blk(bi0, bp);
    const double* Xb1 = A + LAY::blk(bi1, bp);
    double4_t acc0, acc1;
    double aop[4], b0[4], b1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { aop[r] = -Xa[frag<LAY>(r, lane)]; b0[r] = Xb0[frag<LAY>(r, lane)]; b1[r] = Xb1[frag<LAY>(r, lane)]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc0[r] = C0[frag<LAY>(r, lane)]; acc1[r] = C1[frag<LAY>(r, lane)]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[r], b0[r], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[r], b1[r], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { C0[frag<LAY>(r, lane)] = acc0[r]; C1[frag<LAY>(r, lane)] = acc1[r]; }
}

// Up to MB blocks of column bj at once, left-looking: block (i0 + b * istep, bj) -= sum over kb < nk of X(i, kb) X(bj, kb)^T, the operand of
// bj shared, the blocks' chains interleaved, block column kb + 1's operands in flight behind block column kb's products (a lone wavefront
// has nothing else to cover an LDS round trip or a matrix operation's latency with).
template <class LAY, int MB>
__device__ __forceinline__ void lds_update_multi_left(double* __restrict__ A, int i0, int istep, int nb, int bj, int nk, int lane)
{
    double4_t acc[MB];
    double aop[2][4], bop[2][MB][4];
#pragma unroll
    for (int b = 0; b < MB; ++b)
        if (b < nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[b][r] = A[LAY::blk(i0 + b * istep, bj) + frag<LAY>(r, lane)];
        }
    auto fetch = [&](int kb, int slot) {
#pragma unroll
        for (int r = 0; r < 4; ++r) aop[slot][r] = -A[LAY::blk(bj, kb) + frag<LAY>(r, lane)];
#pragma unroll
        for (int b = 0; b < MB; ++b)
            if (b < nb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) bop[slot][b][r] = A[LAY::blk(i0 + b * istep, kb) + frag<LAY>(r, lane)];
            }
    };
    auto products = [&](int slot) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int b = 0; b < MB; ++b)
                if (b < nb) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[slot][r], bop[slot][b][r], acc[b], 0, 0, 0);
    };
    fetch(0, 0);
    for (int kb = 0; kb < nk; kb += 2) {
        if (kb + 1 < nk) fetch(kb + 1, 1);
        products(0);
        if (kb + 1 < nk) {
            if (kb + 2 < nk) fetch(kb + 2, 0);
            products(1);
        }
    }
#pragma unroll
    for (int b = 0; b < MB; ++b)
        if (b < nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) A[LAY::blk(i0 + b * istep, bj) + frag<LAY>(r, lane)] = acc[b][r];
        }
}

// Who touches what, by phase (every pair of a writer and a reader of one location has a barrier between them):
//   recurrence phase   wavefronts 0, 1 READ block (s, s) and their payload blocks of column s (wavefront 0 also its slot of Li) at the start and
//                      WRITE the payload blocks / the slot at the end -- each its own; L_ss itself is stored at once only when wavefront 0 is
//                      alone in the recurrence, otherwise behind barrier A (wavefront 1 may still be reading the block);
//                      the helpers READ columns k < s and READ / WRITE their own blocks of column s + 1; the publisher READS column s - 1
//                      and the other slot of Li;
//   product phase      every wavefront READS column s (written before barrier A) and READS / WRITES its own blocks of column s + 1; one
//                      wavefront refills the other slot of Li with the identity (read by nobody until barrier B); wavefront 0 stores L_ss.
// PUBLISH as above; the progress word reaches base + c once block column c AND the inverse of block c are in memory
// (the strips behind the chain need exactly those, trsm_strip_phased), raised an in-tile iteration after the stores were issued.
// NW = 4 or 8 wavefronts call it (tid 0 .. 64 NW - 1).  A v_mfma_f64_16x16x4 holds its SIMD's matrix pipe for 64 cycles, so what the
// helpers can sum beside a 3 300-cycle recurrence is bounded by the number of SIMDs they run on: with four wavefronts two (three from
// iteration 4 on) share the left-looking sums and the publishing and are late at the barrier in iterations 1-4; with eight, six do --
// a block each -- two of them on the matrix pipes of the recurrence wavefronts' own SIMDs, which the recurrence never uses.
template <bool PARTIAL, class LAY, int PUBLISH = 0, int NW = 4>
__device__ __forceinline__ bool potrf_tile_rows(double* __restrict__ A, double* __restrict__ Li, double* __restrict__ Linv_k, int tid, int nblk = NBLK,
                                             TilePublish pub = TilePublish{})
{
    static_assert(PUBLISH == 0 || PUBLISH == 1 || PUBLISH == 4, "publish forms: 0 plain, 1 write-through inverses, 4 phased");
    static_assert(NW == 4 || NW == 8, "four or eight wavefronts");
    const int NBK = PARTIAL ? nblk : NBLK;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l = lane & 15, g = lane >> 4;
    const int li_off = (int)(Li - A);
    bool failed = false;
    double a_ss[NB];                                       // the pivot block's rows (wavefronts 0, 1)
    // the identity the inverse of block 0 grows from (slot 0; slot (s + 1) & 1 is refilled in iteration s, below)
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int e = lane * 4 + q; Li[e] = (e >> 4) == (e & 15) ? 1.0 : 0.0; }
    }
    // wavefront 3: the inverse of block c (slot c & 1 of Li) and the blocks of column c below the diagonal, out of LDS to memory (4 + 4 (NBK - 1 - c)
    // stores per lane); issued at the top of iteration c + 1, beside the recurrence.
    auto publish = [&](int c) {
        const double* Lc = Li + (c & 1) * NB * NB;
        if (PUBLISH) {
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = Lc[lane * 4 + q];
#pragma unroll
            for (int q = 0; q < 4; ++q) store_through(Linv_k + c * NB * NB + lane * 4 + q, v[q]);
        } else {
            const double4_t v = *reinterpret_cast<const double4_t*>(Lc + lane * 4);
            *reinterpret_cast<double4_t*>(Linv_k + c * NB * NB + lane * 4) = v;
        }
        if (PUBLISH == 4) {
            // (two blocks' reads in flight, then their stores: behind an atomic store the compiler waits for every LDS read it issues)
            for (int i = c + 1; i < NBK; i += 2) {
                const bool two = i + 1 < NBK;
                const double* B0 = A + LAY::blk(i, c);
                const double* B1 = A + LAY::blk(two ? i + 1 : i, c);
                double* G0 = pub.Lpub + (size_t)(i * (i - 1) / 2 + c) * NB * NB;
                double* G1 = pub.Lpub + (size_t)((i + 1) * i / 2 + c) * NB * NB;
                double v0[4], v1[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v0[r] = B0[frag<LAY>(r, lane)]; v1[r] = B1[frag<LAY>(r, lane)]; }
#pragma unroll
                for (int r = 0; r < 4; ++r) store_through(G0 + (4 * r + (lane >> 4)) * NB + (lane & 15), v0[r]);
                if (two) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) store_through(G1 + (4 * r + (lane >> 4)) * NB + (lane & 15), v1[r]);
                }
            }
        }
    };
    // PUBLISH == 4, the end of wavefront 3's share of iteration s >= 2 (it has time until wavefront 0 reaches the barrier): column s - 2 went
    // out a whole iteration and most of this one ago (~3 us; a write-through store is acknowledged after ~2) -- wait for everything but
    // the `newer` stores issued in this iteration (a wavefront's stores complete in order) and raise the progress word.
    auto raise = [&](int c, int newer) {
        switch (newer >> 2) {
            case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
        if (lane == 0) __hip_atomic_store(pub.progress, pub.base + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    for (int s = 0; s < NBK; ++s) {
        const int nstr = NBK - 1 - s;                      // strips below block s
        TILE_STAMP(0);
        // ---- the recurrence (wavefronts 0, 1) beside the left-looking sums of block column s + 1 over k < s (the others)
        const int nrec = nstr > 3 ? 2 : 1;                 // wavefronts inside the recurrence
        if (wave < nrec) {
            const int p = 4 * wave + g;                    // payload of this 16-lane row: 0 strip 0, 1 the identity, p >= 2 strip p - 1
            const int t = p == 0 ? 0 : p - 1;
            // (a row without a strip of its own repeats a payload of its wavefront -- the identity on wavefront 0, strip 3 on wavefront 1:
            // same bits to the same place from the same instruction)
            const bool strip = wave == 1 || (p != 1 && t < nstr);
            const int tt = (wave == 1 && t >= nstr) ? 3 : t;
            const int poff = strip ? LAY::blk(s + 1 + tt, s) + l : li_off + (s & 1) * NB * NB + l;
            const int pst = strip ? LAY::PITCH : NB;
            // L_ss goes into LDS at once only when wavefront 0 is alone in the recurrence: wavefront 1 reads the same block at ITS start, and
            // nothing but the usual pace of two wavefronts says that it has done so 3 000 cycles later (beside other workgroups on its
            // SIMD it may not have: seen as 1.5 results in 1 000 changing under four concurrent handles, tools/soak.py)
            const bool f = factor_block16_rows<LAY::PITCH>(A, LAY::blk(s, s), poff, pst, lane, wave == 0 && nrec == 1, a_ss);
            if (wave == 0) failed |= f;
        } else {
            if (wave == 3 && s >= 1) publish(s - 1);             // column s - 1 and the inverse of block s - 1 go out beside the recurrence
            if (s >= 1 && nstr > 0) {
                // helpers in the order they are dealt blocks: the publisher late, the wavefronts that share a SIMD with the recurrence last
                const int nh = NW - nrec;
                // (eight wavefronts: ranks 0, 1 on the two SIMDs the recurrence does not run on, the publisher -- wavefront 3 -- fourth, the
                // two that share a SIMD with a recurrence wavefront last: 2, 7, 6, 3, 4, 5 measured 35 590 cycles per tile, 2, 6, 7, 3, 4, 5
                // -- the first two on ONE SIMD's matrix pipe -- 36 070, 2, 3, 4, 5, 6, 7 36 720)
                const int hr = NW == 4 ? wave - nrec : (wave == 2 ? 0 : wave == 7 ? 1 : wave == 6 ? 2 : wave == 3 ? 3 : wave == 4 ? 4 : wave == 5 ? 5 : 6);
                const int first = s + 1 + hr;
                const int nb = first < NBK ? (NBK - 1 - first) / nh + 1 : 0;
                if (nb > 0) lds_update_multi_left<LAY, (NW == 4 ? 3 : 1)>(A, first, nh, nb, s + 1, s, lane);
            }
            if (PUBLISH == 4 && wave == 3 && s >= 2) raise(s - 2, 4 + 4 * (NBK - s));
        }
        TILE_STAMP(1);
        tile_barrier<PUBLISH != 0>();                      // (A) column s of L and the inverse of block s are in LDS
        TILE_STAMP(2);
        if (nstr == 0) {
            if (wave == 3) publish(s);                     // the last inverse: nothing left to hide it behind (the caller waits for the stores)
            break;
        }
        // ---- the k = s term of block column s + 1, dealt over the four wavefronts (wavefront 0: the next diagonal block)
        {
            // (wavefront 0, the first into the next recurrence, takes the diagonal block alone)
            const int i0 = s + 1 + wave, i1 = (wave == 0 || NW == 8) ? NBK : i0 + 3;
            if (i1 < NBK) lds_update_tile2<LAY>(A, i0, i1, s + 1, s, lane);
            else if (i0 < NBK) lds_update_tile<LAY>(A, i0, s + 1, s, lane);
        }
        if (wave == 0 && nrec == 2 && lane < NB) {         // L_ss itself, held back while wavefront 1 could still be reading the block (nothing in this phase reads it)
            double* Bss = A + LAY::blk(s, s);
#pragma unroll
            for (int c = 0; c < NB; ++c) Bss[c * LAY::PITCH + l] = a_ss[c];
        }
        if (wave == (NW == 8 ? 7 : 2)) {                   // the identity for the inverse of block s + 1 (slot (s + 1) & 1: the inverse of block s - 1 left it in iteration s - 1)
            double* Ln = Li + ((s + 1) & 1) * NB * NB;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int e = lane * 4 + q; Ln[e] = (e >> 4) == (e & 15) ? 1.0 : 0.0; }
        }
        TILE_STAMP(3);
        tile_barrier<PUBLISH != 0>();                      // (B) block column s + 1 stands at k = s
        TILE_STAMP(4);
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
template <int NT = 256>
__device__ __forceinline__ void load_tile_packed_wt(double* __restrict__ dst, const double* __restrict__ T, int ld, int tid)
{
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(T), 0, (int)((size_t)TILE * ld * sizeof(double)), 0x00020000);
    constexpr int PIECES = PACKED_TILE_DOUBLES / 2, BATCH = 9;
#pragma unroll
    for (int b0 = 0; b0 < PIECES / NT; b0 += BATCH) {
        u32x4_t v[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = (b0 + u) * NT + tid, t = e >> 7, w = e & 127;
            int rb, cb;
            block_of_index(t, rb, cb);
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((size_t)(cb * NB + (w >> 3)) * ld + rb * NB + 2 * (w & 7)) * sizeof(double)), 0, 16);
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int e = (b0 + u) * NT + tid;
            *reinterpret_cast<u32x4_t*>(dst + 2 * e) = v[u];
        }
    }
}
template <int NT = 256>
__device__ __forceinline__ void store_tile_packed_wt(double* __restrict__ T, const double* __restrict__ A, int ld, int tid)
{
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(T, 0, (int)((size_t)TILE * ld * sizeof(double)), 0x00020000);
#pragma unroll 6
    for (int b0 = 0; b0 < PACKED_TILE_DOUBLES / 2 / NT; ++b0) {
        const int e = b0 * NT + tid, t = e >> 7, w = e & 127;
        int rb, cb;
        block_of_index(t, rb, cb);
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4_t*>(A + 2 * e), rsrc,
                                               (int)(((size_t)(cb * NB + (w >> 3)) * ld + rb * NB + 2 * (w & 7)) * sizeof(double)), 0, 16);
    }
}

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

// ---------------------------------------------------------------------------------------------
// trailing update:  C -= L(rows, panels) L(cols, panels)^T  for a block of C held by ONE wavefront in accumulator layout.
// Operands go straight from global memory to registers in MFMA fragment shape (16 consecutive rows x 4 panel columns per load:
// four 128-byte segments), a ring of NBUF chunks of KSTEPS x 4 panel columns ahead of the products; no LDS, no barriers, every
// wavefront independent.  The MFMA "M" index runs over the block's COLUMNS and "N" over its ROWS so that the accumulator's
// lane & 15 direction is the memory-contiguous one.  The accumulators START as the C block and the panel enters negated,
//     acc[a][b][r] = C(row0 + 16 b + (lane & 15), col0 + 16 a + (lane >> 4) + 4 r),      acc -= a_k b_k   for k ascending,
// so an element's value depends only on the ORDER of its panel columns -- never on how a schedule groups them into tasks: one
// task per panel (the column-by-column launches), several panels in one task (the task graph, chol_dag.hip) and any split of a
// tile into blocks give the same bits.  (Rounds 2-4 added C at the end instead -- its load hidden behind the last chunk, 4 %
// faster per task -- which ties the bits to the grouping.)
// Panels k0 .. k1 - 1 are consecutive columns of S, so the ring simply runs on across panel boundaries.
// THROUGH: operands through L1-bypassing loads -- for data handed over inside a launch (MI355X_MICROARCH.md, visibility).
// ---------------------------------------------------------------------------------------------
template <int SUBM, int SUBN, bool THROUGH>
__device__ __forceinline__ void load_c_block(const double* __restrict__ S, int ld, int row0, int col0, int lane, double4_t (&acc)[SUBM][SUBN])
{
    const double* C = S + (size_t)(col0 + (lane >> 4)) * ld + row0 + (lane & 15);
#pragma unroll
    for (int a = 0; a < SUBM; ++a)
#pragma unroll
        for (int b = 0; b < SUBN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = THROUGH ? load_through(C + (size_t)(a * 16 + 4 * r) * ld + b * 16) : C[(size_t)(a * 16 + 4 * r) * ld + b * 16];
}
template <int SUBM, int SUBN, bool THROUGH>
__device__ __forceinline__ void store_c_block(double* __restrict__ S, int ld, int row0, int col0, int lane, const double4_t (&acc)[SUBM][SUBN])
{
    double* C = S + (size_t)(col0 + (lane >> 4)) * ld + row0 + (lane & 15);
#pragma unroll
    for (int a = 0; a < SUBM; ++a)
#pragma unroll
        for (int b = 0; b < SUBN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (THROUGH) store_through(C + (size_t)(a * 16 + 4 * r) * ld + b * 16, acc[a][b][r]);
                else C[(size_t)(a * 16 + 4 * r) * ld + b * 16] = acc[a][b][r];
            }
}
template <int SUBM, int SUBN, int KSTEPS, int NBUF, bool THROUGH>
__device__ __forceinline__ void panel_update(const double* __restrict__ S, int ld, int k0, int k1, int row0, int col0, int lane, double4_t (&acc)[SUBM][SUBN])
{
    const double* Pn = S + (size_t)(k0 * TILE + (lane >> 4)) * ld + row0 + (lane & 15);
    const double* Pm = S + (size_t)(k0 * TILE + (lane >> 4)) * ld + col0 + (lane & 15);
    constexpr int NCH = TILE / (4 * KSTEPS);             // chunks per panel (a multiple of NBUF)
    static_assert(NCH % NBUF == 0, "the ring is unrolled NBUF chunks at a time");
    const int nch = (k1 - k0) * NCH;
    double av[NBUF][KSTEPS][SUBM], bv[NBUF][KSTEPS][SUBN];
    auto load_chunk = [&](int buf, int ch) {
#pragma unroll
        for (int s4 = 0; s4 < KSTEPS; ++s4) {
            const size_t off = (size_t)(ch * 4 * KSTEPS + s4 * 4) * ld;
#pragma unroll
            for (int q = 0; q < SUBM; ++q) av[buf][s4][q] = THROUGH ? load_through(Pm + off + q * 16) : Pm[off + q * 16];
#pragma unroll
            for (int q = 0; q < SUBN; ++q) bv[buf][s4][q] = THROUGH ? load_through(Pn + off + q * 16) : Pn[off + q * 16];
        }
    };
#pragma unroll
    for (int p = 0; p < NBUF - 1; ++p) load_chunk(p, p);
    for (int ch0 = 0; ch0 < nch; ch0 += NBUF) {
#pragma unroll
        for (int u = 0; u < NBUF; ++u) {
            const int ch = ch0 + u;
            // The prefetch is UNCONDITIONAL (past the end it fetches the last chunk again, into a buffer nobody reads): behind a condition
            // the two paths into the products carry different numbers of loads in flight and the compiler must wait for the lower one --
            // s_waitcnt vmcnt(0) in front of every chunk, i.e. for the prefetch it has just issued (rounds 1-5: one exposed trip to memory
            // per two chunks; a lone wavefront ran its products at ~55 % of the pipe's rate, two per SIMD covered for each other to ~85 %).
            if (NBUF == 1) load_chunk(0, ch);
            else load_chunk((u + NBUF - 1) % NBUF, ch + NBUF - 1 < nch ? ch + NBUF - 1 : nch - 1);
            // the panel enters negated on the side with fewer registers ((-a) b and a (-b) are the same product, bit for bit)
#pragma unroll
            for (int s4 = 0; s4 < KSTEPS; ++s4)
#pragma unroll
                for (int a = 0; a < SUBM; ++a)
#pragma unroll
                    for (int b = 0; b < SUBN; ++b)
                        acc[a][b] = SUBN < SUBM ? __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][s4][a], -bv[u][s4][b], acc[a][b], 0, 0, 0)
                                                : __builtin_amdgcn_mfma_f64_16x16x4f64(-av[u][s4][a], bv[u][s4][b], acc[a][b], 0, 0, 0);
        }
    }
}

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

// PHASED strip of the merged panel solve (round 4): the factoring workgroup publishes block column c of L_kk (and the block inverses)
// as it goes (potrf_tile_rows<.., 4>: write-through stores, progress words raised one in-tile iteration later, when the stores have long
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

}  // namespace chol
}  // namespace mage
