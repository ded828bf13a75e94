// chol_dag.hip -- the tiled Cholesky factorisation + forward substitution of chol_kernels.hip as ONE persistent launch.
//
// Why: the column-by-column schedule pays max(chain, bulk) PER COLUMN -- while the trailing update is large the chain of diagonal
// tiles (factor tile k+1 <- update it <- strips of column k <- factor tile k) idles behind it, and once the update is small the GPU
// idles behind the chain (profiles/HISTORY.md, rounds 2-4; tools/chol_dag_sim.py prices the alternatives).  Here the whole
// factorisation is a graph of tile tasks taken from a STATIC list by whichever team is free, behind dependency counters:
//
//   chain workgroup   (block 0, wavefronts 0-3; the other four leave): for k = 0 .. nt-1: wait for the nine arrivals at diagonal tile
//                     k, pull it into LDS, factor it (potrf_tile_lds), write L_kk and its block inverses through, raise `fact`.
//   worker teams      every other workgroup = two teams of four wavefronts (512 threads, one workgroup per compute unit, <= 256
//                     registers: two wavefronts per SIMD, the shape in which the trailing update runs best).  A team's leader takes
//                     the next task of the list (one returning atomic), polls the task's dependency words, and the team runs it:
//       STRIP(i,k,s)    one 16-row strip of the panel solve L_ik = S_ik L_kk^-T by the four wavefronts (each owns two 16-column
//                       blocks, Y_j handed round through LDS), then y_i[rows] -= L_ik[rows] y_k -- the forward substitution rides
//                       on the strips, in the order the column-by-column launches apply it;
//       YSOLVE(k)       y_k = L_kk^-1 y_k, the same strip code on the rhs row;
//       UPDATE(i,j,u,k0,nk)  block u of tile (i,j) -= L_i,k L_j,k^T for the nk panels k0 .. k0+nk-1 IN ONE TASK: the block is read and
//                       written once however many panels it absorbs (a right-looking launch moves C once per panel: 15x the
//                       compulsory traffic, r04_chol_pmc.txt), and the operand ring runs on across panels.  Half tiles (two units
//                       per tile) while tiles are plenty, quarter tiles in the last columns;
//       DIAG(j,p)       the LAST panel (j-1) of diagonal tile j, split over nine teams (36 blocks of 16 x 16, one per wavefront) because
//                       it sits on the chain; the ninth arrival releases the chain workgroup.
//
// The list is the start order of a list-scheduling simulation of that graph (build_schedule below: measured task costs, earliest
// column first, a unit takes every panel that is available when its turn comes -- far tiles accumulate panels while near ones are
// served, which is where the large-K tasks come from).  Every dependency of a task sits earlier in the list or is a chain task
// whose own dependencies do (check_schedule proves it per list): teams that take tasks in list order therefore cannot deadlock,
// whatever the real timings are; a wrong cost only costs time.
//
// Numbers: every element accumulates its panel columns in ascending order into an accumulator that STARTS as the element
// (chol_device.h, panel_update), strips and in-tile factorisation are the column-by-column launches' code: the factor and y have
// the same bits under ANY schedule, this one and the launches (tests/test_chol_gpu.py compares the solution's bits).
//
// Visibility: everything a task hands on is stored THROUGH (sc1) and counted after the stores are acknowledged; a consumer polls
// the counter (relaxed, agent scope), its leader issues ONE agent-scope acquire (L1 invalidate) before the team starts, and reads
// plainly (MI355X_MICROARCH.md: recipe R1).  Waits are bounded; one that runs out raises `abort` + *stall = 4, every team leaves at
// its next task, and the host runs the trial again column by column (chol_report_stall).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <queue>
#include <tuple>
#include <vector>
#include "chol_dag.h"
#include "chol_device.h"
#include "mage_common.h"

namespace mage {
using namespace chol;
namespace {

// ---- state words (ints, zeroed before every launch), relative to CholWorkspace::sync + 8
constexpr int D_HEAD = 0;       // cursor of the task list
constexpr int D_FACT = 1;       // tile columns whose diagonal tile is factored and in memory
constexpr int D_YSOL = 2;       // rhs rows solved
constexpr int D_ABORT = 3;      // a bounded wait ran out somewhere: everybody leaves
constexpr int D_PROG = 4;       // phased hand-off: 8 tile + block columns of the tile being factored that are published
constexpr int D_DARR = 16;      // nt ints: arrivals of the split last panel at diagonal tile j
__host__ __device__ inline int tri(int i, int j) { return i * (i + 1) / 2 + j; }
__host__ __device__ inline int d_stripc(int nt) { return D_DARR + nt; }                          // tri(i, k): strips of tile (i, k) done (8 = L_ik complete)
__host__ __device__ inline int d_usum(int nt) { return d_stripc(nt) + nt * (nt + 1) / 2; }       // tri(i, j): panels applied, summed over the tile's units
__host__ __device__ inline int d_uprog(int nt) { return d_usum(nt) + nt * (nt + 1) / 2; }        // 4 tri(i, j) + u: panels applied to unit u
inline int dag_state_ints(int nt) { return d_uprog(nt) + 4 * (nt * (nt + 1) / 2); }

// ---- tasks: type | i | j | unit | k0 | nk in one 64-bit word
enum : unsigned { T_END = 0, T_STRIP = 1, T_HALF = 2, T_QUARTER = 3, T_DIAG = 4, T_YSOLVE = 5 };
__host__ __device__ inline unsigned long long task_word(unsigned type, unsigned i, unsigned j, unsigned unit, unsigned k0, unsigned nk)
{
    return (unsigned long long)type | ((unsigned long long)i << 8) | ((unsigned long long)j << 16) | ((unsigned long long)unit << 24) |
           ((unsigned long long)k0 << 32) | ((unsigned long long)nk << 40);
}
__host__ __device__ inline unsigned t_type(unsigned long long w) { return (unsigned)(w & 0xff); }
__host__ __device__ inline int t_i(unsigned long long w) { return (int)((w >> 8) & 0xff); }
__host__ __device__ inline int t_j(unsigned long long w) { return (int)((w >> 16) & 0xff); }
__host__ __device__ inline int t_unit(unsigned long long w) { return (int)((w >> 24) & 0xff); }
__host__ __device__ inline int t_k0(unsigned long long w) { return (int)((w >> 32) & 0xff); }
__host__ __device__ inline int t_nk(unsigned long long w) { return (int)((w >> 40) & 0xff); }
// units of tile (i, j): halves (2) in the columns before quarter_from, quarters behind (the diagonal tile's upper-right quarter does not exist)
__host__ __device__ inline int tile_units(int i, int j, int quarter_from) { return j >= quarter_from ? (i == j ? 3 : 4) : 2; }

constexpr int DAG_THREADS = 512;
constexpr int DAG_SPIN_LIMIT = 1 << 20;

typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(1))) int global_int;

struct DagArgs {
    double* S; double* y; double* x; double* Linv; double* Lpub; double* ok; double* stall;
    int* st;                               // state words
    const unsigned long long* tasks;
    int ld, nt, n_tasks, quarter_from;
    long long* trace;                      // development (tools/chol_test.hip built with -DDAG_TRACE): per task 4 stamps of the 100 MHz clock, behind them 2 per tile column of the chain
};

#ifdef DAG_TRACE
#define DAG_STAMP(slot) do { if (a.trace) a.trace[slot] = wall_clock64(); } while (0)
#else
#define DAG_STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ int ld_word(const int* p) { return __hip_atomic_load((const global_int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_word(int* p, int v) { __hip_atomic_store((global_int*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void add_word(int* p, int v) { __hip_atomic_fetch_add((global_int*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one lane: wait until *w >= target; false when the wait ran out or somebody else's did
__device__ __forceinline__ bool poll_ge(const int* w, int target, const int* abort)
{
    for (int spins = 0;; ++spins) {
        if (ld_word(w) >= target) return true;
        if ((spins & 63) == 63 && ld_word(abort)) return false;
        if (spins >= DAG_SPIN_LIMIT) return false;
        __builtin_amdgcn_s_sleep(2);
    }
}

// ---- a team = four wavefronts of a 512-thread workgroup; its barrier is an LDS counter (the two teams of a workgroup run
// independently, so the hardware barrier is not theirs to use)
struct Team {
    lds_int* cnt;
    int phase, lane, tw;
};
__device__ __forceinline__ void team_sync(Team& t)
{
    t.phase += 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (t.lane == 0) {
        __hip_atomic_fetch_add(t.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(t.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < t.phase) __builtin_amdgcn_s_sleep(0);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// One 16-row strip of the panel solve by the four wavefronts of a team.  A strip is a chain of 176 matrix-core operations, but only 8
// of a step's operations are on the chain; wavefront W owns block columns {W, 7 - W} (9 products each): the owner of column j forms
// Y_j = Linv_jj acc_j, leaves it in LDS (accumulator layout = B operand of the updates), one team barrier, every wavefront applies Y_j
// to the columns it owns -- the next step's column first.  Per block column the products meet the accumulator in the order of the
// one-wavefront strip (trsm_strip): bit-identical.  L_kk and the block inverses come through L1-bypassing loads, the result goes
// through to memory.  ysh: 8 x 256 doubles of LDS; on return (after the caller's barrier) it holds the whole strip, element (row n,
// column c) at ysh[(c >> 4) * 256 + (c & 15) * 16 + n].
// ---------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void dag_strip_body(double* __restrict__ base, size_t cstride, bool live, const double* __restrict__ S, int ld, int k,
                                               const double* __restrict__ Linv_k, Team& t, double* __restrict__ ysh)
{
    constexpr int C0 = W, C1 = NBLK - 1 - W;
    const int lane = t.lane;
    double4_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        acc0[r] = live ? base[(size_t)(C0 * NB + (lane >> 4) + 4 * r) * cstride] : 0.0;
        acc1[r] = live ? base[(size_t)(C1 * NB + (lane >> 4) + 4 * r) * cstride] : 0.0;
    }
    const double* Lop = S + (size_t)(k * TILE + (lane >> 4)) * ld + (size_t)k * TILE + (lane & 15);
    const double* Lio = Linv_k + (lane & 15) * NB + (lane >> 4);
    double lio0[4], lio1[4], lop0[C0 > 0 ? C0 : 1][4], lop1[C1][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { lio0[r] = load_through(Lio + C0 * NB * NB + 4 * r); lio1[r] = load_through(Lio + C1 * NB * NB + 4 * r); }
#pragma unroll
    for (int j = 0; j < C0; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) lop0[j][r] = -load_through(Lop + (size_t)(j * NB + 4 * r) * ld + C0 * NB);
#pragma unroll
    for (int j = 0; j < C1; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) lop1[j][r] = -load_through(Lop + (size_t)(j * NB + 4 * r) * ld + C1 * NB);
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
        double4_t Yj = { 0, 0, 0, 0 };
        const bool mine = j == C0 || j == C1;
        if (mine) {
            const double4_t a = j == C0 ? acc0 : acc1;
#pragma unroll
            for (int r = 0; r < 4; ++r) Yj = __builtin_amdgcn_mfma_f64_16x16x4f64(j == C0 ? lio0[r] : lio1[r], a[r], Yj, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) ysh[j * 256 + r * 64 + lane] = Yj[r];
            if (live) {
#pragma unroll
                for (int r = 0; r < 4; ++r) store_through(base + (size_t)(j * NB + (lane >> 4) + 4 * r) * cstride, Yj[r]);
            }
        }
        if (j == NBLK - 1) break;
        team_sync(t);
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
__device__ __forceinline__ void dag_strip(double* __restrict__ base, size_t cstride, bool live, const double* __restrict__ S, int ld, int k,
                                          const double* __restrict__ Linv_k, Team& t, double* __restrict__ ysh)
{
    switch (t.tw) {
        case 0: dag_strip_body<0>(base, cstride, live, S, ld, k, Linv_k, t, ysh); break;
        case 1: dag_strip_body<1>(base, cstride, live, S, ld, k, Linv_k, t, ysh); break;
        case 2: dag_strip_body<2>(base, cstride, live, S, ld, k, Linv_k, t, ysh); break;
        default: dag_strip_body<3>(base, cstride, live, S, ld, k, Linv_k, t, ysh); break;
    }
}

#ifndef DAG_UPDATE_INLINE
#define DAG_UPDATE_INLINE __forceinline__
#endif
// One wavefront's block of an update task: read C, absorb panels k0 .. k1 - 1, write C through.  A function of its own (no LDS in it, so
// nothing is lost by the call): inlined beside the strips the three shapes were allocated against them and ~90 registers went to scratch.
template <int SUBM, int SUBN, int KSTEPS, int NBUF>
__device__ DAG_UPDATE_INLINE void update_task(double* __restrict__ S, int ld, int k0, int k1, int row0, int col0, int lane)
{
    double4_t acc[SUBM][SUBN];
    load_c_block<SUBM, SUBN, false>(S, ld, row0, col0, lane, acc);
    panel_update<SUBM, SUBN, KSTEPS, NBUF, false>(S, ld, k0, k1, row0, col0, lane, acc);
    store_c_block<SUBM, SUBN, true>(S, ld, row0, col0, lane, acc);
}

// ---------------------------------------------------------------------------------------------
// the launch
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DAG_THREADS) void k_chol_dag(DagArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: the team state derived from it stays out of the vector registers)
    const int nt = a.nt, ld = a.ld;
    int* const st = a.st;
    int* const stripc = st + d_stripc(nt);
    int* const usum = st + d_usum(nt);
    int* const uprog = st + d_uprog(nt);
    const size_t linv_stride = (size_t)NBLK * NB * NB;

    if (blockIdx.x == 0) {
        // ================= the chain workgroup =================
        if (wave >= 4) {
            // the other four wavefronts open the solve -- x pre-filled with the sentinel the backward substitution polls for -- and leave
            // (the hardware barrier below then counts the four that stay)
            unsigned long long* xf = reinterpret_cast<unsigned long long*>(a.x);
            for (int i = tid - 256; i < nt * TILE; i += 256) xf[i] = X_SENTINEL;
            return;
        }
        if (tid == 0) { *a.ok = 1.0; *a.stall = 0.0; }
        double* A = sm;                            // LayPacked: the 36 lower blocks
        double* Li = sm + PACKED_TILE_DOUBLES;     // 2 x (16 x 16)
        int* bail = reinterpret_cast<int*>(sm + PACKED_TILE_DOUBLES + 2 * NB * NB);
        if (tid == 0) *bail = 0;
        for (int k = 0; k < nt; ++k) {
            if (k > 0) {
                if (tid == 0 && !poll_ge(st + D_DARR + k, NDIAG, st + D_ABORT)) {
                    if (!ld_word(st + D_ABORT)) { *a.stall = 4.0; st_word(st + D_ABORT, 1); }
                    *bail = 1;
                }
                __syncthreads();
                if (*bail) return;
            }
            double* T = a.S + (size_t)(k * TILE) * ld + (size_t)k * TILE;
            if (tid == 0) DAG_STAMP(4 * (size_t)a.n_tasks + 2 * k);
            load_tile_packed_wt(A, T, ld, tid);
            __syncthreads();
            const bool failed = potrf_tile_lds<false, LayPacked, 1>(A, Li, a.Linv + (size_t)k * linv_stride, tid);
            store_tile_packed_wt(T, A, ld, tid);
            if (tid == 0 && failed) *a.ok = 0.0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) { st_word(st + D_FACT, k + 1); DAG_STAMP(4 * (size_t)a.n_tasks + 2 * k + 1); }
        }
        return;
    }

    // ================= worker teams =================
    const int team = wave >> 2;
    Team t;
    lds_int* const ctl = (lds_int*)(sm + 2 * (NBLK * NB * NB)) + team * 8;          // [0] barrier counter, [2..3] mailbox
    double* const ysh = sm + team * (NBLK * NB * NB);
    double* const yks = sm + 2 * (NBLK * NB * NB) + 8 + team * TILE;                 // y_k for the strips' rhs rows
    t.cnt = ctl; t.phase = 0; t.lane = lane; t.tw = wave & 3;
    if ((tid & 255) == 0) { ctl[0] = 0; ctl[2] = 0; ctl[3] = 0; ctl[4] = 0; }
    __syncthreads();                                           // the only workgroup-wide barrier: both teams are still together here
    while (true) {
        // (the lane index is made opaque once per task: otherwise every address the four strip bodies and the update shapes derive from it
        // is hoisted out of this loop, ~70 registers held across all roles and spilled)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        t.lane = ln;
        const bool leader = t.tw == 0 && ln == 0;
        if (leader) {
            unsigned long long w = 0;
            const int id = __hip_atomic_fetch_add((global_int*)(st + D_HEAD), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (id < a.n_tasks && !ld_word(st + D_ABORT)) w = a.tasks[id];
            if (w) DAG_STAMP(4 * (size_t)id);
            bool ready = true;
            const int i = t_i(w), j = t_j(w), u = t_unit(w), k0 = t_k0(w), nk = t_nk(w);
            switch (t_type(w)) {
                case T_STRIP:      // tile (i, k = j) has absorbed its j panels; L_jj is in memory
                    ready = poll_ge(st + D_FACT, j + 1, st + D_ABORT) &&
                            (j == 0 || poll_ge(usum + tri(i, j), tile_units(i, j, a.quarter_from) * j, st + D_ABORT));
                    break;
                case T_YSOLVE:     // y_j has been updated by every strip of tile row j; L_jj is in memory
                    ready = poll_ge(st + D_FACT, j + 1, st + D_ABORT) && (j == 0 || poll_ge(stripc + tri(j, j - 1), NBLK, st + D_ABORT));
                    break;
                case T_HALF:
                case T_QUARTER:    // the unit stands at panel k0; the strips of the last panel it takes are complete (they imply the earlier ones)
                    ready = poll_ge(uprog + 4 * tri(i, j) + u, k0, st + D_ABORT) && poll_ge(stripc + tri(i, k0 + nk - 1), NBLK, st + D_ABORT) &&
                            (i == j || poll_ge(stripc + tri(j, k0 + nk - 1), NBLK, st + D_ABORT));
                    break;
                case T_DIAG:       // the diagonal tile has absorbed panels 0 .. j-2; L_{j,j-1} is complete
                    ready = poll_ge(stripc + tri(j, j - 1), NBLK, st + D_ABORT) &&
                            poll_ge(usum + tri(j, j), tile_units(j, j, a.quarter_from) * (j - 1), st + D_ABORT);
                    break;
                default: break;
            }
            if (!ready) {
                if (!ld_word(st + D_ABORT)) { *a.stall = 4.0; st_word(st + D_ABORT, 1); }
                w = 0;
            }
            if (w) DAG_STAMP(4 * (size_t)id + 1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // what the producers stored through is read plainly from here on
            if (w) DAG_STAMP(4 * (size_t)id + 2);
            ctl[2] = (int)(unsigned)w; ctl[3] = (int)(unsigned)(w >> 32); ctl[4] = id;
        }
        team_sync(t);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane(ctl[2]), hi = (unsigned)__builtin_amdgcn_readfirstlane(ctl[3]);
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;
        const unsigned type = t_type(w);
        if (type == T_END) break;
        const int i = t_i(w), j = t_j(w), u = t_unit(w), k0 = t_k0(w), nk = t_nk(w);
        if (type == T_HALF) {
            // half u of tile (i, j): 128 rows x 64 columns, a wavefront 64 x 32
            if (!(i == j && u == 1 && (t.tw & 1) == 0)) {         // diagonal tile: rows 0-63 of columns 64-127 lie above the diagonal
                const int row0 = i * TILE + (t.tw & 1) * 64, col0 = j * TILE + u * 64 + (t.tw >> 1) * 32;
                update_task<2, 4, 4, 2>(a.S, ld, k0, k0 + nk, row0, col0, ln);
            }
        } else if (type == T_QUARTER) {
            const int row0 = i * TILE + (u & 1) * 64 + (t.tw & 1) * 32, col0 = j * TILE + ((u >> 1) & 1) * 64 + (t.tw >> 1) * 32;
            update_task<2, 2, 8, 2>(a.S, ld, k0, k0 + nk, row0, col0, ln);
        } else if (type == T_DIAG) {
            int bi, bj;
            tile_of_index(u * 4 + t.tw, bi, bj);
            const int row0 = j * TILE + bi * NB, col0 = j * TILE + bj * NB;
            update_task<1, 1, 32, 1>(a.S, ld, j - 1, j, row0, col0, ln);
        } else if (type == T_YSOLVE) {
            dag_strip(a.y + (size_t)j * TILE, 1, (ln & 15) == 0, a.S, ld, j, a.Linv + (size_t)j * linv_stride, t, ysh);
        } else {      // T_STRIP: strip u of tile (i, k = j), then y_i[16 rows] -= L_ik[rows] y_k
            dag_strip(a.S + (size_t)(j * TILE) * ld + (size_t)i * TILE + u * NB + (ln & 15), (size_t)ld, true, a.S, ld, j,
                      a.Linv + (size_t)j * linv_stride, t, ysh);
            team_sync(t);
            if (t.tw == 0) {
                bool solved = true;
                if (ln == 0) solved = poll_ge(st + D_YSOL, j + 1, st + D_ABORT);
                solved = __builtin_amdgcn_readfirstlane((int)solved) != 0;
                if (!solved) {
                    if (ln == 0 && !ld_word(st + D_ABORT)) { *a.stall = 4.0; st_word(st + D_ABORT, 1); }
                } else {
                    // y_k in ONE trip (two values per lane, parked in LDS), then one chain of 128 products per row on lanes 0-15 -- the order in
                    // which the column-by-column launches add them.  (First form: 128 L1-bypassing loads in the chain, eight at a time: 25 us.)
                    const double* yk = a.y + (size_t)j * TILE;
                    const double y0 = load_through(yk + 2 * ln), y1 = load_through(yk + 2 * ln + 1);
                    double* yi = a.y + (size_t)i * TILE + u * NB + (ln & 15);
                    const double yold = load_through(yi);
                    yks[2 * ln] = y0; yks[2 * ln + 1] = y1;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    if (ln < NB) {
                        double accv = 0;
#pragma unroll 16
                        for (int c = 0; c < TILE; ++c) accv = __builtin_fma(ysh[(c >> 4) * 256 + (c & 15) * 16 + ln], yks[c], accv);
                        store_through(yi, yold - accv);
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wavefront's stores are in memory
        team_sync(t);
        if (leader) {
            DAG_STAMP(4 * (size_t)ctl[4] + 3);
            switch (type) {
                case T_STRIP: add_word(stripc + tri(i, j), 1); break;
                case T_YSOLVE: st_word(st + D_YSOL, j + 1); break;
                case T_DIAG: add_word(st + D_DARR + j, 1); break;
                default: st_word(uprog + 4 * tri(i, j) + u, k0 + nk); add_word(usum + tri(i, j), nk); break;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The schedule: list scheduling of the task graph in simulated time (costs in us, round-4 / round-5 measurements).
// ---------------------------------------------------------------------------------------------
struct SimCosts {                         // us, from the stamps of tools/_bin/chol_test_trace 6016 (profiles/r05_dag_trace_*.txt)
    double potrf = 21.2;                  // tile -> LDS, in-tile factorisation, factor -> memory, flag
    double strip = 8.0;                   // one strip by four wavefronts + its rhs rows
    double ysolve = 8.0;
    double diag = 6.0, gather = 1.5;      // one ninth of the split panel; ninth arrival -> chain workgroup sees it
    double hop = 1.5;                     // a predecessor on another compute unit becomes visible to a poll
    double half0 = 10.5, half1 = 16.3;    // half-tile task: fixed (C in and out, ramp, drain, hand-off) + per panel
    double quarter0 = 4.5, quarter1 = 10.1;
};

struct Ev {
    double t; long seq; int kind, a, b, c, d;
    bool operator>(const Ev& o) const { return t != o.t ? t > o.t : seq > o.seq; }
};
enum { EV_SREADY, EV_SDONE, EV_DREADY, EV_DDONE, EV_UDONE, EV_POTRF, EV_YREADY, EV_YDONE, EV_UREADY };

std::vector<unsigned long long> build_schedule(int nt, int n_teams, int gmax, int& quarter_from)
{
    const SimCosts C;
    // quarter tiles from the first column whose trailing matrix no longer offers a half-tile task per team
    quarter_from = nt;
    for (int j = 1; j < nt; ++j) { const int m = nt - j; if (m * (m + 1) < n_teams) { quarter_from = j; break; } }
    const int ntri = nt * (nt + 1) / 2;
    std::vector<int> strip_cnt(ntri, 0), availc(ntri, 0), nxt(4 * ntri, 0), darr(nt, 0);
    std::vector<char> busy(4 * ntri, 0), queued(4 * ntri, 0), tile_done(ntri, 0), strips_out(ntri, 0), diag_out(nt, 0), y_out(nt, 0);
    std::vector<double> potrf_end(nt, 0.0);
    int fact = 0;
    std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> events;
    long seq = 0;
    auto post = [&](double t, int kind, int a2 = 0, int b = 0, int c = 0, int d = 0) { events.push(Ev{ t, seq++, kind, a2, b, c, d }); };
    // ready queues: (key..., payload)
    typedef std::tuple<int, int, long, int, int, int> Key;      // (column, row, seq, i, j, unit)
    std::priority_queue<Key, std::vector<Key>, std::greater<Key>> ready_units, ready_strips;
    std::vector<std::pair<int, int>> ready_diag;
    std::vector<int> ready_y;
    std::vector<unsigned long long> list;
    int free_teams = n_teams;
    double now = 0;
    auto units_of = [&](int i, int j) { return tile_units(i, j, quarter_from); };
    auto unit_id = [&](int i, int j, int idx) { return (j >= quarter_from && i == j && idx == 2) ? 3 : idx; };      // the diagonal tile's quarters are 0, 1, 3
    auto limit_of = [&](int i, int j) { return i == j ? j - 1 : j; };       // panels the regular units apply (the diagonal tile's last one is split)
    auto tile_complete = [&](int i, int j) {
        for (int x = 0; x < units_of(i, j); ++x) if (nxt[4 * tri(i, j) + unit_id(i, j, x)] < limit_of(i, j)) return false;
        return true;
    };
    auto release_strips = [&](int i, int k, double t) {
        if (strips_out[tri(i, k)]) return;
        strips_out[tri(i, k)] = 1;
        for (int s = 0; s < NBLK; ++s) post(t, EV_SREADY, i, k, s);
    };
    auto consider = [&](int i, int j, double t) {
        int& av = availc[tri(i, j)];
        while (av < j && strip_cnt[tri(i, av)] == NBLK && strip_cnt[tri(j, av)] == NBLK) ++av;
        const int lim = std::min(av, limit_of(i, j));
        for (int x = 0; x < units_of(i, j); ++x) {
            const int u = unit_id(i, j, x), id = 4 * tri(i, j) + u;
            if (!busy[id] && !queued[id] && nxt[id] < lim) { queued[id] = 1; post(t, EV_UREADY, i, j, u); }
        }
        if (i == j && !diag_out[j] && av >= j && tile_complete(j, j)) {
            diag_out[j] = 1;
            for (int p = 0; p < NDIAG; ++p) post(t, EV_DREADY, j, p);
        }
    };
    auto start_potrf = [&](int k, double t) {
        potrf_end[k] = std::max(t, k > 0 ? potrf_end[k - 1] : 0.0) + C.potrf;
        post(potrf_end[k], EV_POTRF, k);
    };
    start_potrf(0, 0.0);
    while (true) {
        while (free_teams > 0 && (!ready_diag.empty() || !ready_y.empty() || !ready_strips.empty() || !ready_units.empty())) {
            if (!ready_diag.empty()) {
                const auto [j, p] = ready_diag.back(); ready_diag.pop_back();
                list.push_back(task_word(T_DIAG, j, j, p, j - 1, 1));
                --free_teams; post(now + C.diag, EV_DDONE, j, p);
                continue;
            }
            if (!ready_y.empty()) {
                const int k = ready_y.back(); ready_y.pop_back();
                list.push_back(task_word(T_YSOLVE, k, k, 0, 0, 0));
                --free_teams; post(now + C.ysolve, EV_YDONE, k);
                continue;
            }
            if (!ready_strips.empty()) {
                const Key key = ready_strips.top(); ready_strips.pop();
                const int i = std::get<3>(key), k = std::get<4>(key), s = std::get<5>(key);
                list.push_back(task_word(T_STRIP, i, k, s, 0, 0));
                --free_teams; post(now + C.strip, EV_SDONE, i, k, s);
                continue;
            }
            const Key key = ready_units.top(); ready_units.pop();
            const int i = std::get<3>(key), j = std::get<4>(key), u = std::get<5>(key), id = 4 * tri(i, j) + u;
            queued[id] = 0;
            const int lim = std::min(availc[tri(i, j)], limit_of(i, j));
            const int n = std::min(lim - nxt[id], gmax);
            if (n <= 0) continue;
            const bool quarter = j >= quarter_from;
            list.push_back(task_word(quarter ? T_QUARTER : T_HALF, i, j, u, nxt[id], n));
            busy[id] = 1; --free_teams;
            post(now + (quarter ? C.quarter0 + n * C.quarter1 : C.half0 + n * C.half1), EV_UDONE, i, j, u, n);
        }
        if (events.empty()) break;
        now = events.top().t;
        while (!events.empty() && events.top().t <= now) {      // everything that happens at this instant, then dispatch
        const Ev e = events.top(); events.pop();
        switch (e.kind) {
            case EV_POTRF: {
                const int k = e.a;
                fact = k + 1;
                if (!y_out[k] && (k == 0 || strip_cnt[tri(k, k - 1)] == NBLK)) { y_out[k] = 1; post(now + C.hop, EV_YREADY, k); }      // (always: the split panel needed L_{k,k-1} complete)
                for (int i = k + 1; i < nt; ++i) if (k == 0 || tile_done[tri(i, k)]) release_strips(i, k, now + C.hop);
                break;
            }
            case EV_SREADY: ready_strips.push(Key{ e.b, e.a, seq++, e.a, e.b, e.c }); break;
            case EV_YREADY: ready_y.push_back(e.a); break;
            case EV_DREADY: ready_diag.push_back({ e.a, e.b }); break;
            case EV_UREADY: ready_units.push(Key{ e.b, e.a, seq++, e.a, e.b, e.c }); break;
            case EV_YDONE: ++free_teams; break;
            case EV_DDONE:
                ++free_teams;
                if (++darr[e.a] == NDIAG) start_potrf(e.a, now + C.hop + C.gather);
                break;
            case EV_SDONE: {
                ++free_teams;
                const int i = e.a, k = e.b;
                if (++strip_cnt[tri(i, k)] == NBLK) {
                    for (int j = k + 1; j <= i; ++j) consider(i, j, now + C.hop);
                    for (int i2 = i + 1; i2 < nt; ++i2) consider(i2, i, now + C.hop);
                    if (i == k + 1 && fact >= i + 1 && !y_out[i]) { y_out[i] = 1; post(now + C.hop, EV_YREADY, i); }
                }
                break;
            }
            case EV_UDONE: {
                ++free_teams;
                const int i = e.a, j = e.b, u = e.c, id = 4 * tri(i, j) + u;
                busy[id] = 0; nxt[id] += e.d;
                if (i != j && !tile_done[tri(i, j)] && tile_complete(i, j)) {
                    tile_done[tri(i, j)] = 1;
                    if (fact >= j + 1) release_strips(i, j, now + C.hop);
                }
                consider(i, j, now);
                break;
            }
            default: break;
        }
        }
    }
    return list;
}

// Every dependency of the task at position p is produced by tasks at positions < p, or by a chain task (potrf(k)) whose own
// dependencies are: replay the list with instantaneous tasks and check each task's wait conditions at its position.  Also checks
// that the list is complete (every strip, every panel of every unit, every split panel, every rhs row exactly once).
bool check_schedule(const std::vector<unsigned long long>& list, int nt, int quarter_from)
{
    const int ntri = nt * (nt + 1) / 2;
    std::vector<int> stripc(ntri, 0), usum(ntri, 0), uprog(4 * ntri, 0), darr(nt, 0);
    int fact = 1, ysol = 0;          // potrf(0) depends on nothing
    auto advance_chain = [&] { while (fact < nt && darr[fact] == NDIAG) ++fact; };
    for (unsigned long long w : list) {
        const int i = t_i(w), j = t_j(w), u = t_unit(w), k0 = t_k0(w), nk = t_nk(w);
        switch (t_type(w)) {
            case T_STRIP:
                if (fact < j + 1 || (j > 0 && usum[tri(i, j)] < tile_units(i, j, quarter_from) * j)) return false;
                if (ysol < j + 1) return false;        // (its rhs rows wait for y_j inside the task: the solve must sit earlier in the list too)
                ++stripc[tri(i, j)];
                break;
            case T_YSOLVE:
                if (fact < j + 1 || ysol != j || (j > 0 && stripc[tri(j, j - 1)] < NBLK)) return false;
                ysol = j + 1;
                break;
            case T_HALF:
            case T_QUARTER:
                if (nk < 1 || uprog[4 * tri(i, j) + u] != k0 || stripc[tri(i, k0 + nk - 1)] < NBLK || (i != j && stripc[tri(j, k0 + nk - 1)] < NBLK)) return false;
                if (k0 + nk > (i == j ? j - 1 : j)) return false;
                uprog[4 * tri(i, j) + u] = k0 + nk; usum[tri(i, j)] += nk;
                break;
            case T_DIAG:
                if (stripc[tri(j, j - 1)] < NBLK || usum[tri(j, j)] < tile_units(j, j, quarter_from) * (j - 1)) return false;
                ++darr[j]; advance_chain();
                break;
            default: return false;
        }
    }
    if (fact != nt || ysol != nt) return false;
    for (int i = 1; i < nt; ++i)
        for (int j = 0; j < i; ++j) if (stripc[tri(i, j)] != NBLK) return false;
    for (int i = 1; i < nt; ++i)
        for (int j = 1; j <= i; ++j) if (usum[tri(i, j)] != tile_units(i, j, quarter_from) * (i == j ? j - 1 : j)) return false;
    return true;
}

struct DagSchedule { unsigned long long* d_tasks = nullptr; int n_tasks = 0, quarter_from = 0; bool ok = false; };
std::mutex g_sched_mutex;
std::map<std::pair<int, int>, DagSchedule> g_sched;      // (device, nt)
int g_dag_n_cu = 256;

int dag_min_tiles()
{
    static const int v = [] { const char* e = std::getenv("MAGE_CHOL_DAG_MIN_TILES"); return e ? std::atoi(e) : 8; }();
    return v;
}
int dag_fuse_max()
{
    static const int v = [] { const char* e = std::getenv("MAGE_CHOL_DAG_FUSE"); return e ? std::max(1, std::min(64, std::atoi(e))) : 8; }();
    return v;
}

const DagSchedule* get_schedule(int nt)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_sched_mutex);
    auto it = g_sched.find({ dev, nt });
    if (it == g_sched.end()) {
        DagSchedule s;
        std::vector<unsigned long long> list = build_schedule(nt, 2 * (g_dag_n_cu - 1), dag_fuse_max(), s.quarter_from);
        if (check_schedule(list, nt, s.quarter_from) && hipMalloc(&s.d_tasks, list.size() * sizeof(unsigned long long)) == hipSuccess) {
            if (hipMemcpy(s.d_tasks, list.data(), list.size() * sizeof(unsigned long long), hipMemcpyHostToDevice) == hipSuccess) { s.n_tasks = (int)list.size(); s.ok = true; }
            else { (void)hipFree(s.d_tasks); s.d_tasks = nullptr; }
        }
        if (!s.ok) (void)hipGetLastError();
        it = g_sched.emplace(std::make_pair(dev, nt), s).first;
    }
    return it->second.ok ? &it->second : nullptr;
}

}  // namespace

size_t chol_dag_sync_ints(int nt) { return (size_t)dag_state_ints(nt); }

void chol_dag_init_device(int n_cu)
{
    g_dag_n_cu = n_cu;
    const size_t lds = ((size_t)PACKED_TILE_DOUBLES + 2 * NB * NB + 16) * sizeof(double);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_dag), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

bool chol_dag_factor(double* S, double* y, double* x, int n_pad, const CholWorkspace& ws, double* ok, double* stall, hipStream_t st)
{
    const int nt = n_pad / TILE;
    if (nt < dag_min_tiles() || nt > 255 || g_dag_n_cu < 8) return false;
    const DagSchedule* s = get_schedule(nt);
    if (!s) return false;
    int* state = ws.sync + 8;
    if (hipMemsetAsync(state, 0, (size_t)dag_state_ints(nt) * sizeof(int), st) != hipSuccess) { (void)hipGetLastError(); return false; }
    DagArgs a;
    a.S = S; a.y = y; a.x = x; a.Linv = ws.Linv; a.Lpub = ws.Linv + (size_t)nt * NBLK * NB * NB; a.ok = ok; a.stall = stall;
    a.st = state; a.trace = ws.dbg; a.tasks = s->d_tasks; a.ld = n_pad; a.nt = nt; a.n_tasks = s->n_tasks; a.quarter_from = s->quarter_from;
    const size_t lds = ((size_t)PACKED_TILE_DOUBLES + 2 * NB * NB + 16) * sizeof(double);
    hipLaunchKernelGGL(k_chol_dag, dim3(g_dag_n_cu), dim3(DAG_THREADS), lds, st, a);
    if (hipGetLastError() != hipSuccess) return false;
    return true;
}

}  // namespace mage

// Host-only view of the schedule for tests (tests/test_chol_schedule.py): the list for nt tile columns on n_cu compute units, and
// whether check_schedule accepts it.  Returns the list's length (<= cap entries are written), negative when the list fails the check.
MAGE_EXPORT int mage_debug_chol_schedule(int nt, int n_cu, int fuse_max, unsigned long long* out, int cap, int* quarter_from)
{
    if (nt < 2 || nt > 255 || n_cu < 2) return 0;
    int qf = 0;
    std::vector<unsigned long long> list = mage::build_schedule(nt, 2 * (n_cu - 1), fuse_max > 0 ? fuse_max : 8, qf);
    if (quarter_from) *quarter_from = qf;
    for (int i = 0; i < (int)list.size() && i < cap; ++i) out[i] = list[i];
    return mage::check_schedule(list, nt, qf) ? (int)list.size() : -(int)list.size();
}
