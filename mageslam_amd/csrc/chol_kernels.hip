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
#include "chol_device.h"
#include "chol_dag.h"

namespace mage {
using namespace chol;
namespace {

// Once a bounded wait between workgroups of ONE launch has run out in this process (several PROCESSES oversubscribing the GPU: the
// hardware scheduler saves and restores workgroups), every later factorisation uses launches without such waits: chol_report_stall.
std::atomic<bool> g_merge_disabled{ false };

// ---------------------------------------------------------------------------------------------
// Tile 0, stand-alone: LDS-resident, blocked by 16 (potrf_tile_rows).  It also opens the solve: ok = 1, stall = 0, and x pre-filled
// with the sentinel the backward substitution polls for.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_potrf_diag(double* __restrict__ S, int ld, int k, double* __restrict__ Linv_k,
                                                    double* __restrict__ ok, double* __restrict__ stall, unsigned long long* __restrict__ x_fill, int n_fill, unsigned long long* __restrict__ part_fill)
{
    extern __shared__ double sm[];
    double* A = sm;                       // LayPacked: the 36 lower blocks
    double* Li = sm + PACKED_TILE_DOUBLES;   // 2 x (16 x 16): inverse of the current / next diagonal block
    const int tid = threadIdx.x;
    if (tid == 0) { *ok = 1.0; *stall = 0.0; }
    for (int i = tid; i < n_fill; i += 256) { x_fill[i] = X_SENTINEL; part_fill[i] = X_SENTINEL; }      // (the backward solve's partial sums: same protocol)
    double* T = S + (size_t)(k * TILE) * ld + (size_t)k * TILE;
    load_tile_packed(A, T, ld, tid);
    __syncthreads();
    const bool failed = potrf_tile_rows<false, LayPacked>(A, Li, Linv_k, tid);
    store_tile_packed(T, A, ld, tid);
    if (tid == 0 && failed) *ok = 0.0;
}

// the panel solve of tile column k as a launch of its own (column 0; the update-bound columns; every column once merging is off):
// one wavefront per 16-row strip (all 36 operand blocks in ~380 registers), the last workgroup solves the rhs row.  It also zeroes the
// hand-off counters of the launches that follow (flag[0] per launch, all of them at column 0).
__global__ __launch_bounds__(64) void k_trsm_panel_1w(double* __restrict__ S, double* __restrict__ y, int ld, int k, int nt,
                                                      const double* __restrict__ Linv_k, int* __restrict__ flag)
{
    const int lane = threadIdx.x;
    if (blockIdx.x == 0 && lane == 0) { flag[0] = 0; if (k == 0) { flag[1] = 0; flag[2] = 0; flag[3] = 0; flag[4] = 0; } }
    const int n_strips = (nt - k - 1) * NBLK;
    trsm_strip(S, y, ld, k, (int)blockIdx.x, (int)blockIdx.x == n_strips, Linv_k, lane);
}

// ---- hand-offs inside a merged launch.  flag[0]: arrivals at the split diagonal tile; flag[1]: the last tile column whose diagonal
// tile is factored and in memory; flag[2]: parts of first-column tiles written, cumulative over the launches of a factorisation;
// flag[4]: progress inside the tile being factored (phased strips).
// Producers of first-column tiles: every store of the workgroup done, ONE agent-scope release (an L2 write-back), then the count.
// Consumers (the strips, dispatched behind every producer they wait for): relaxed polls, one acquire.
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

// y_i -= L_ik y_k for one tile row (a workgroup), one chain of 128 products per row
__device__ __forceinline__ void rhs_row_update(const double* __restrict__ S, double* __restrict__ y, int ld, int k, int i, int tid)
{
    const double* Lik = S + (size_t)(k * TILE) * ld + (size_t)i * TILE;
    const double* yk = y + (size_t)k * TILE;
    for (int r = tid; r < TILE; r += 256) {
        double acc = 0;
#pragma unroll 8
        for (int c = 0; c < TILE; ++c) acc = __builtin_fma(Lik[(size_t)c * ld + r], yk[c], acc);
        y[(size_t)i * TILE + r] -= acc;
    }
}

// ---------------------------------------------------------------------------------------------
// Trailing update by panel k in the CHAIN-BOUND columns (fewer than ~400 tiles left): one workgroup per compute unit (the phased strip
// keeps ~390 registers).  Grid, in dispatch order (j0 = k + 1, tiles (i, j) with j0 <= j <= i < nt):
//   blocks 0..8     the NEXT diagonal tile (j0, j0): its 36 lower 16x16 blocks, one per wavefront; all write THROUGH to S; blocks 1..8
//                   count up flag[0]; block 0 waits for 8, pulls the tile into LDS, factors it in place (publishing block column by
//                   block column, potrf_tile_rows<.., 4>) and raises flag[1] -- the factorisation of step k + 1 overlaps this update;
//   quarter tiles   every other tile in four 64 x 64 blocks, 32 x 32 per wavefront (a whole tile on one wavefront per SIMD takes
//                   24-44 us with 130-250 of them in flight: in quarters the update ends before the chain does);
//   m blocks        rhs rows y_i -= L_ik y_k; the one of row j0 then solves y_j0 as a strip;
//   MERGED:         (m - 1) * 8 strips of the panel solve of column j0 (wavefront 0 of each block), PHASED against block 0's
//                   factorisation (trsm_strip_phased), behind the first-column tiles of this very launch (flag[2]).
// Progress relies on in-order dispatch: every waiter is dispatched behind the producers it waits for.  Waits are bounded; one that
// runs out sets *stall (1 split diagonal tile, 2 merged panel solve) and the host runs the trial again with MERGED = false.
// ---------------------------------------------------------------------------------------------
template <bool MERGED>
__global__ __launch_bounds__(256) void k_syrk_update(double* __restrict__ S, double* __restrict__ y, int ld, int k, int nt,
                                                     double* __restrict__ Linv_next, double* __restrict__ ok, double* __restrict__ stall, int* __restrict__ flag, int col_target,
                                                     double* __restrict__ Lpub_next)
{
    extern __shared__ double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0 = k + 1, mt = nt - j0;
    const int n_tiles = mt * (mt + 1) / 2;
    const int first_rhs = NDIAG + 4 * (n_tiles - 1);
    const int bid = blockIdx.x;
    if (bid >= first_rhs + mt) {
        // ---- merged panel solve of tile column j0: one strip per workgroup (wavefront 0)
        if (!MERGED || wave != 0) return;
        trsm_strip_phased(S, y, ld, j0, bid - (first_rhs + mt), false, Linv_next, Lpub_next, flag, col_target, stall, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    if (bid >= first_rhs) {
        const int i = k + 1 + (bid - first_rhs);
        rhs_row_update(S, y, ld, k, i, tid);
        if (MERGED && i == j0) {
            // the rhs row of the merged panel solve: y_j0 is this workgroup's own (just updated), L_j0j0 comes from workgroup 0
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (wave == 0) trsm_strip_phased(S, y, ld, j0, 0, true, Linv_next, Lpub_next, flag, 0, stall, lane);
        }
        return;
    }
    if (bid < NDIAG) {
        // 16x16 block u = 4 bid + wave of the lower triangle of the diagonal tile, (bi, bj), bi >= bj
        const int u = bid * 4 + wave;
        int bi, bj;
        tile_of_index(u, bi, bj);
        const int row0 = j0 * TILE + bi * NB, col0 = j0 * TILE + bj * NB;
        double4_t acc[1][1];
        load_c_block<1, 1, false>(S, ld, row0, col0, lane, acc);
        panel_update<1, 1, 32, 1, false>(S, ld, k, k + 1, row0, col0, lane, acc);
        store_c_block<1, 1, true>(S, ld, row0, col0, lane, acc);         // written THROUGH: no release fence below
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (bid != 0) {
            if (tid == 0) __hip_atomic_fetch_add(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        // block 0: wait for the eight others (bounded spin, relaxed polls; the tile comes through sc1 loads: no acquire), pull the tile, factor it
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NDIAG - 1 && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(1);
            if (spins >= (1 << 24)) *stall = 1.0;        // a producer never arrived: reported as a device error (never folded into "not positive definite")
            __hip_atomic_store(flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        double* A = sm;
        double* T = S + (size_t)(j0 * TILE) * ld + (size_t)j0 * TILE;
        load_tile_packed_wt(A, T, ld, tid);
        __syncthreads();
        const bool failed = MERGED ? potrf_tile_rows<false, LayPacked, 4>(A, sm + PACKED_TILE_DOUBLES, Linv_next, tid, NBLK, TilePublish{ Lpub_next, flag + 4, 8 * j0 })
                                   : potrf_tile_rows<false, LayPacked, 0>(A, sm + PACKED_TILE_DOUBLES, Linv_next, tid);
        if (tid == 0 && failed) *ok = 0.0;
        if (MERGED) {
            // the strips read the published blocks and the block inverses, all written through: raise the flag FIRST, the factor itself goes
            // to S afterwards (the updates and the backward solve read it there, launches later)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flag + 1, j0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        store_tile_packed(T, A, ld, tid);
        return;
    }
    // quarter of tile index 1 + q / 4
    const int q = bid - NDIAG;
    int rt, ct;
    tile_of_index(1 + (q >> 2), rt, ct);
    if (rt == ct && (q & 3) == 2) return;                     // diagonal tile: rows 0-63 of columns 64-127 lie above the diagonal
    const int row0 = (j0 + rt) * TILE + (q & 1) * 64 + (wave & 1) * 32;
    const int col0 = (j0 + ct) * TILE + ((q >> 1) & 1) * 64 + (wave >> 1) * 32;
    double4_t acc[2][2];
    load_c_block<2, 2, false>(S, ld, row0, col0, lane, acc);
    panel_update<2, 2, 8, 2, false>(S, ld, k, k + 1, row0, col0, lane, acc);
    store_c_block<2, 2, false>(S, ld, row0, col0, lane, acc);
    if (MERGED && ct == 0) publish_column_part(flag, tid);
}

// ---------------------------------------------------------------------------------------------
// Trailing update by panel k in the UPDATE-BOUND columns: half tiles (128 rows x 64 columns per workgroup, 64 x 32 per wavefront), two
// workgroups per compute unit (~200 registers, 78 KB of LDS), the two halves of a tile on one XCD; the tiles of a last round that
// fills at most half of the task slots go in quarters (q_tiles).  Updated blocks are stored THROUGH (nothing of them is left dirty for
// the kernel boundary).  The next diagonal tile is split over nine workgroups and factored by workgroup 0 as above; the panel solve
// of column k + 1 is a launch of its own behind this one (four ways of merging it were measured slower: profiles/HISTORY.md).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_syrk_update2(double* __restrict__ S, double* __restrict__ y, int ld, int k, int nt,
                                                         double* __restrict__ Linv_next, double* __restrict__ ok, double* __restrict__ stall, int* __restrict__ flag, int q_tiles)
{
    extern __shared__ double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0 = k + 1, mt = nt - j0;
    const int n_tiles = mt * (mt + 1) / 2;
    const int n_half_tiles = n_tiles - 1 - q_tiles;               // tile indices 1 .. n_half_tiles in halves
    const int n_task_wgs = 16 * ((n_half_tiles + 7) / 8);         // two workgroups (column halves) each, in groups of eight tiles
    const int first_q4 = NDIAG + n_task_wgs;
    const int first_rhs = first_q4 + 4 * q_tiles;
    const int bid = blockIdx.x;
    if (bid >= first_rhs) {
        const int i = k + 1 + (bid - first_rhs);
        if (i < nt) rhs_row_update(S, y, ld, k, i, tid);
        return;
    }
    if (bid < NDIAG) {
        const int u = bid * 4 + wave;
        int bi, bj;
        tile_of_index(u, bi, bj);
        const int row0 = j0 * TILE + bi * NB, col0 = j0 * TILE + bj * NB;
        double4_t acc[1][1];
        load_c_block<1, 1, false>(S, ld, row0, col0, lane, acc);
        panel_update<1, 1, 32, 1, false>(S, ld, k, k + 1, row0, col0, lane, acc);
        store_c_block<1, 1, false>(S, ld, row0, col0, lane, acc);
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
            if (spins >= (1 << 24)) *stall = 1.0;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        double* A = sm;
        double* T = S + (size_t)(j0 * TILE) * ld + (size_t)j0 * TILE;
        load_tile_packed(A, T, ld, tid);
        __syncthreads();
        const bool failed = potrf_tile_rows<false, LayPacked>(A, sm + PACKED_TILE_DOUBLES, Linv_next, tid);
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
        double4_t acc[2][2];
        load_c_block<2, 2, false>(S, ld, row0, col0, lane, acc);
        panel_update<2, 2, 8, 2, false>(S, ld, k, k + 1, row0, col0, lane, acc);
        store_c_block<2, 2, true>(S, ld, row0, col0, lane, acc);
        return;
    }
    // half of a tile: workgroups 16 g + t and 16 g + 8 + t serve tile 8 g + t (workgroups go to the eight XCDs round-robin: the two
    // halves of a tile meet in one L2).  Placement only: any order gives the same numbers.
    const int q0 = bid - NDIAG;
    const int tile_i = 8 * (q0 >> 4) + (q0 & 7);
    if (tile_i >= n_half_tiles) return;
    const int h = (q0 >> 3) & 1;
    int rt, ct;
    tile_of_index(1 + tile_i, rt, ct);
    if (rt == ct && h && (wave & 1) == 0) return;                 // diagonal tile: rows 0-63 of columns 64-127 lie above the diagonal
    const int row0 = (j0 + rt) * TILE + (wave & 1) * 64, col0 = (j0 + ct) * TILE + h * 64 + (wave >> 1) * 32;
    double4_t acc[2][4];
    load_c_block<2, 4, false>(S, ld, row0, col0, lane, acc);
    panel_update<2, 4, 4, 2, false>(S, ld, k, k + 1, row0, col0, lane, acc);
    store_c_block<2, 4, true>(S, ld, row0, col0, lane, acc);
}

// ---------------------------------------------------------------------------------------------
// backward substitution  L^T x = y  as ONE persistent launch: tile column j is owned by a pair of workgroups (below).  The closer keeps L_jj (LDS) and
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


// TWO workgroups per tile column (round 5).  A column consumes one 128-KB tile per arriving x_k, and fetching it into the LDS buffer can
// only start when the buffer's previous tile has moved to registers: ~2.9 us per arrival against the 1.9 us a hop takes (CHOL_DBG=1
// tools/_bin/chol_test: every second hop was waiting for the consumer, not for x).  So the arrivals of column j alternate between two
// workgroups: the CLOSER (role 1: k = j + 1, j + 3, ... -- it takes the last arrival, applies L_jj^-T and publishes x_j) and the HELPER
// (role 0: k = j + 2, j + 4, ..., starting from zero), which hands its partial sums over through `part` (sentinel-polled like x) one
// hop before they are needed.  Each has two hops per tile.  The helper's sums join the closer's just before its last product.
__global__ __launch_bounds__(256) void k_bsolve_persist(const double* __restrict__ S, const double* __restrict__ y, double* __restrict__ x,
                                                        int ld, int nt, const double* __restrict__ Linv, double* __restrict__ stall, long long* __restrict__ dbg,
                                                        double* __restrict__ part, const int* __restrict__ kmax)
{
    extern __shared__ double sm[];
    constexpr int LDB = TILE + 2;
    double* T = sm;                        // first L_jj^-T, then the off-diagonal tile staged for the NEXT arrival; column-major, pitch LDB
    __shared__ double ys[TILE], xk[TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = nt - 1 - (int)(blockIdx.x >> 1);              // the last tile column is dispatched first
    const int closer = (int)(blockIdx.x & 1);                   // (the helper first: its closer waits for it)
    const int k_last = kmax ? kmax[j] : nt - 1;                 // the last tile row of this column inside the skyline (a dense system: every row)
    const int k_first = k_last - ((k_last - (j + 2 - closer)) & 1);       // the largest k <= k_last of this role's parity (k = j + 2 - closer, + 2, ...)
    const bool has_work = k_first > j;
    if (!closer && !has_work) return;                           // no arrival for the helper (the closer knows: k_last - j < 2)
    if (tid < TILE) ys[tid] = closer ? y[(size_t)j * TILE + tid] : 0.0;
    // thread (c, half) owns rows half * 64 .. +63 of column c of the current off-diagonal tile, and columns half * 64 .. of row c of the inverse
    const int c = tid >> 1, half = tid & 1;
    double minv[64], tile[64];
    if (closer) {
        // The tile's inverse by the panel solve's strip code on the rows of the identity (two strips per wavefront, ~10 us, while every
        // column but the last two is waiting anyway): the end of a hop is then one product from registers, not eight substitution steps.
        trsm_strip<true>(const_cast<double*>(S), nullptr, ld, j, wave, false, Linv + (size_t)j * NBLK * NB * NB, lane, T, LDB);
        trsm_strip<true>(const_cast<double*>(S), nullptr, ld, j, wave + 4, false, Linv + (size_t)j * NBLK * NB * NB, lane, T, LDB);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 64; ++q) minv[q] = T[(half * 64 + q) * LDB + c];
    }
    __syncthreads();                       // T is free
    // The tiles travel global -> LDS by the load-to-LDS path of wavefronts 2-3 (one fully coalesced 1-KB column per instruction, no
    // registers), a whole arrival ahead; every thread takes its share from LDS into registers; and wavefronts 0-1, which poll, never have
    // a tile load in flight (a wavefront's loads return in order: a poll issued behind a tile fetch cannot return before it has landed).
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
    if (has_work) {
        if (wave >= 2) { stage(k_first); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __syncthreads();
        take();
        __syncthreads();
        if (wave >= 2 && k_first - 2 > j) stage(k_first - 2);
    }
    if (dbg && closer && tid == 0) { dbg[j * 4 + 0] = wall_clock64(); dbg[j * 4 + 1] = dbg[j * 4 + 0]; }
    auto poll_values = [&](const double* from) {      // 128 values that replace the sentinel when their producer stores them
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(from + tid);
        unsigned long long v;
        int spins = 0;
        while ((v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == X_SENTINEL && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
        if (v == X_SENTINEL) *stall = 3.0;           // the producer never published: reported as a device error
        return __longlong_as_double((long long)v);
    };
    for (int k = k_first; k > j; k -= 2) {
        if (k == j + 1 && dbg && tid == 0) dbg[j * 4 + 2] = wall_clock64();
        // before the closer's last arrival: the helper's sums (its last tile is one hop older: they are here or about to be, and this
        // wait is the one that has to be sat out anyway)
        if (closer && k == j + 1 && k_last - j >= 2 && tid < TILE) ys[tid] += poll_values(part + (size_t)j * TILE);
        if (tid < TILE) xk[tid] = poll_values(x + (size_t)k * TILE);
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
        if (k - 2 > j) {
            if (wave >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the role's next tile has landed in T
            __syncthreads();
            take();
            __syncthreads();
            if (wave >= 2 && k - 4 > j) stage(k - 4);
        } else __syncthreads();
    }
    if (!closer) {
        // the helper's share of y_j - sum_k L_kj^T x_k (a negative sum): to the closer
        if (tid < TILE) __hip_atomic_store(reinterpret_cast<unsigned long long*>(part + (size_t)j * TILE + tid), (unsigned long long)__double_as_longlong(ys[tid]),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
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
// ONE launch of one workgroup -- the leading ceil(n / 16) blocks of the tile factored in LDS (potrf_tile_rows<true>), then both
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
    const bool failed = potrf_tile_rows<true, LayLDC>(A, Li, Linv, tid, nblk);
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
// that the phased strips read
size_t chol_workspace_doubles(int n_pad) { return (size_t)(n_pad / TILE) * (NBLK * NB * NB + LPUB_TILE_DOUBLES + TILE); }      // block inverses, published L_kk copies, the backward solve's partial sums
size_t chol_sync_ints(int n_pad) { return 8 + chol_dag_sync_ints(n_pad / TILE); }

int g_n_cu = 256;        // compute units of the device the library was initialised on (gfx950: 256)

bool chol_merge_fallback_active() { return g_merge_disabled.load(std::memory_order_relaxed); }

// The host saw *stall = code: 1 split diagonal tile, 2 merged panel solve, 3 backward solve, 4 the task-graph launch.  Codes 2 and 4
// name waits that a schedule WITHOUT them avoids: from then on this process factors column by column with separate panel-solve launches.
void chol_report_stall(int code)
{
    if (code == 2 || code == 4) g_merge_disabled.store(true, std::memory_order_relaxed);
}

// Kernels that need more than the default dynamic-LDS limit must be opted in once per device (function attributes
// are per device); called from mage_ba_create after hipSetDevice.
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
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_update<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_diag);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_update<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_diag);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_update2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_diag);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_bsolve_persist), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_panel);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_solve<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_small_solve<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small);
    chol_dag_init_device(g_n_cu);
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

// Two schedules of the same arithmetic (same operations in the same order per element: the same bits):
//   * ONE launch for the whole factorisation + forward substitution, its tasks taken from a static list by resident teams behind
//     dependency counters, so that the chain of diagonal tiles runs ahead of the trailing updates (chol_dag.hip) -- the default from
//     CHOL_DAG_MIN_TILES tile columns on;
//   * column by column: step k = the trailing update by column k, which factors the diagonal tile of column k + 1 inside (workgroup 0)
//     -- and, in the chain-bound columns, also SOLVES column k + 1's panel inside (phased strips at the end of the grid); the
//     update-bound columns are followed by a panel-solve launch.  Small systems, MAGE_CHOL_COLUMN_LAUNCHES=1, and every factorisation
//     after a bounded wait of the other schedule has run out in this process (then without merged strips: no in-launch wait is left
//     but the split diagonal tile's, whose producers are the first nine workgroups of the grid).
// The backward substitution is k_bsolve_persist either way.
// A launch the runtime refused (an LDS opt-in that did not take, a device in a bad state) must not leave `ok` / `stall` stale and the
// factor -- or an x still full of the backward solve's sentinel -- behind a MAGE_OK: poison both so that the caller's read-back reports a
// device error.  Run behind the LAST launch of either schedule (the task-graph path used to return in front of it).
static void poison_on_launch_error(double* ok, double* stall, hipStream_t st)
{
    if (hipGetLastError() == hipSuccess) return;
    static const double bad[2] = { 0.0, 9.0 };          // (static: the copy is asynchronous)
    (void)hipMemcpyAsync(ok, &bad[0], sizeof(double), hipMemcpyHostToDevice, st);
    (void)hipMemcpyAsync(stall, &bad[1], sizeof(double), hipMemcpyHostToDevice, st);
}

void chol_factor_solve(double* S, double* y, double* x, int n_pad, const CholWorkspace& ws, double* ok, hipStream_t st)
{
    const int nt = n_pad / TILE;
    const size_t lds_diag = ((size_t)PACKED_TILE_DOUBLES + 2 * NB * NB) * sizeof(double);
    const size_t lds_panel = (size_t)TILE * (TILE + 2) * sizeof(double);
    const size_t linv_stride = (size_t)NBLK * NB * NB;
    double* stall = ws.stall ? ws.stall : ok + 1;     // callers without a slot of their own pass a two-element ok
    double* const bs_part = ws.Linv + (size_t)nt * (linv_stride + LPUB_TILE_DOUBLES);      // the backward solve's partial sums, behind the published L_kk copies
    static const bool column_launches = std::getenv("MAGE_CHOL_COLUMN_LAUNCHES") != nullptr;
    static const bool merge_env_off = std::getenv("MAGE_CHOL_NO_MERGED_TRSM") != nullptr;      // (tests: the fall-back schedule on demand)
    const bool stalled_before = g_merge_disabled.load(std::memory_order_relaxed);
    const bool merge_off = merge_env_off || stalled_before;
    const int* kmax_dev = nullptr;
    if (!column_launches && !stalled_before && chol_dag_factor(S, y, x, n_pad, ws, ok, stall, st, &kmax_dev)) {
#ifdef DAG_TRACE
        long long* const bs_dbg = nullptr;            // (the trace build hands ws.dbg to the task-graph launch)
#else
        long long* const bs_dbg = ws.dbg;
#endif
        hipLaunchKernelGGL(k_bsolve_persist, dim3(2 * nt), dim3(256), lds_panel, st, S, y, x, n_pad, nt, ws.Linv, stall, bs_dbg, bs_part, kmax_dev);
        poison_on_launch_error(ok, stall, st);
        return;
    }
    int* flag = ws.sync;
    hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(256), lds_diag, st, S, n_pad, 0, ws.Linv, ok, stall, reinterpret_cast<unsigned long long*>(x), n_pad, reinterpret_cast<unsigned long long*>(bs_part));
    hipLaunchKernelGGL(k_trsm_panel_1w, dim3((nt - 1) * NBLK + 1), dim3(64), 0, st, S, y, n_pad, 0, nt, ws.Linv, flag);
    int col_total = 0;
    for (int k = 0; k + 1 < nt; ++k) {
        const int m = nt - k - 1;             // tile rows below panel k = tile rows of the trailing matrix
        const int n_tiles = m * (m + 1) / 2;
        double* const Linv_next = ws.Linv + (size_t)(k + 1) * linv_stride;
        bool merged = false;
        // While the update is what takes the time (more than ~1.5 rounds of tiles) the half-tile form runs two workgroups per compute
        // unit; once the chain (diagonal update, hand-off, in-tile factorisation) is what takes the time, the workgroup that factors must
        // not share its compute unit: the quarter-tile form, one workgroup per unit.
        if (n_tiles >= 400) {
            // tiles of the last, partial round of half-tile tasks go in quarters when that round fills at most half of the task slots
            const int rem_tiles = (n_tiles - 1) % g_n_cu;
            const int q_tiles = (rem_tiles > 0 && rem_tiles * 2 <= g_n_cu) ? rem_tiles : 0;
            hipLaunchKernelGGL(k_syrk_update2, dim3(NDIAG + 16 * ((n_tiles - 1 - q_tiles + 7) / 8) + 4 * q_tiles + m), dim3(256), lds_diag, st, S, y, n_pad, k, nt,
                               Linv_next, ok, stall, flag, q_tiles);
        } else {
            merged = !merge_off;
            if (merged) col_total += 4 * (m - 1);      // workgroups of this launch that write a part of tile column k + 1 below the diagonal tile
            const dim3 grid(NDIAG + 4 * (n_tiles - 1) + m + (merged ? (m - 1) * NBLK : 0));
            double* const Lpub_next = ws.Linv + (size_t)nt * linv_stride + (size_t)(k + 1) * LPUB_TILE_DOUBLES;
            if (merged) hipLaunchKernelGGL(k_syrk_update<true>, grid, dim3(256), lds_diag, st, S, y, n_pad, k, nt, Linv_next, ok, stall, flag, col_total, Lpub_next);
            else hipLaunchKernelGGL(k_syrk_update<false>, grid, dim3(256), lds_diag, st, S, y, n_pad, k, nt, Linv_next, ok, stall, flag, col_total, Lpub_next);
        }
        if (!merged) hipLaunchKernelGGL(k_trsm_panel_1w, dim3((m - 1) * NBLK + 1), dim3(64), 0, st, S, y, n_pad, k + 1, nt, Linv_next, flag);
    }
    // backward substitution: one persistent launch (needs every workgroup resident: nt <= 256 compute units)
    hipLaunchKernelGGL(k_bsolve_persist, dim3(2 * nt), dim3(256), lds_panel, st, S, y, x, n_pad, nt, ws.Linv, stall, ws.dbg, bs_part, static_cast<const int*>(nullptr));
    poison_on_launch_error(ok, stall, st);
}

}  // namespace mage
