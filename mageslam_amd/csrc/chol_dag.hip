// chol_dag.hip -- the tiled Cholesky factorisation + forward substitution of chol_kernels.hip as ONE persistent launch.
//
// Why: the column-by-column schedule pays max(chain, bulk) PER COLUMN -- while the trailing update is large the chain of diagonal
// tiles (factor tile k+1 <- update it <- strips of column k <- factor tile k) idles behind it, and once the update is small the GPU
// idles behind the chain (profiles/HISTORY.md, rounds 2-4; tools/chol_dag_sim.py prices the alternatives).  Here the whole
// factorisation is a graph of tile tasks taken from a STATIC list by whichever team is free, behind dependency counters:
//
//   chain workgroup   (block 0, all eight wavefronts: two carry the pivot recurrence, six sum beside it): for k = 0 .. nt-1: wait for the nine arrivals at diagonal tile
//                     k, pull it into LDS, factor it (potrf_tile_rows), write L_kk and its block inverses through, raise `fact`.
//   worker teams      every other workgroup = two teams of four wavefronts (512 threads, one workgroup per compute unit, <= 256
//                     registers: two wavefronts per SIMD, the shape in which the trailing update runs best).  A team's leader takes
//                     the next task of the list (one returning atomic), polls the task's dependency words, and the team runs it:
//       STRIPS(i,k,part)  the panel solve L_ik = S_ik L_kk^-T of a whole tile (or half of it, for the rows next to the chain): the team parks
//                       L_kk's sub-diagonal blocks and block inverses in LDS ONCE (operand layout, 72 KB) and every wavefront then runs
//                       the one-wavefront strip of chol_device.h on its 16-row strips with operands from LDS -- no hand-off inside;
//       RHS(i,k0,nk)    y_i -= L_ik y_k for nk panels, YSOLVE(k)  y_k = L_kk^-1 y_k: the forward substitution as its own chain of small
//                       tasks, in the order the column-by-column launches apply it;
//       UPDATE(i,j,u,k0,nk)  block u of tile (i,j) -= L_i,k L_j,k^T for the nk panels k0 .. k0+nk-1 IN ONE TASK: the block is read and
//                       written once however many panels it absorbs (a right-looking launch moves C once per panel: 15x the
//                       compulsory traffic, r04_chol_pmc.txt), and the operand ring runs on across panels.  Half tiles (two units
//                       per tile) while tiles are plenty, quarter tiles in the last columns;
//       DIAG(j,p)       the LAST panel (j-1) of diagonal tile j, split over nine teams (36 blocks of 16 x 16, one per wavefront) because
//                       it sits on the chain; the ninth arrival releases the chain workgroup.
//
// Teams are in eight GROUPS (workgroup index mod 8 = the XCD it runs on, for speed only) with a task list each: the update tasks of a
// 4 x 2 block of tiles belong to one group, so that the tasks an XCD runs side by side share their panel operands in its L2 (with one
// list for all, every task's operands came from beyond the L2: 18 half-panels / us machine-wide against the launches' 24,
// profiles/r05_dag_trace_v1_6016.txt).
// A ninth list, the EXPRESS list, holds the small tasks next to the chain -- the split panels, the strips and completing quarters of the two
// rows below the diagonal, the rhs solves -- and ten workgroups serve nothing else: in a group's list such a task waited up to 30 us for a
// team behind long updates (profiles/r05_dag_trace_*.txt).
// The lists are the start order of a list-scheduling simulation of that graph (build_schedule below: measured task costs, earliest
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
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <queue>
#include <tuple>
#include <type_traits>
#include <vector>
#include "chol_dag.h"
#include "chol_device.h"
#include "mage_common.h"

namespace mage {
using namespace chol;
namespace {

// ---- state words (ints, zeroed before every launch), relative to CholWorkspace::sync + 8
constexpr int D_FACT = 1;       // tile columns whose diagonal tile is factored and in memory
constexpr int D_YSOL = 2;       // rhs rows solved
constexpr int D_ABORT = 3;      // a bounded wait ran out somewhere: everybody leaves
constexpr int D_INJECT = 5;     // test hook (mage_debug_chol_inject_stall): the chain workgroup treats its first wait as run out
constexpr int D_PROG = 4;       // phased hand-off: 8 tile + block columns of the tile being factored that are in memory
constexpr int D_HEADS = 16;     // cursor of group g's task list at D_HEADS + 16 g (a cache line each)
constexpr int N_GROUPS = 8;
constexpr int N_LISTS = N_GROUPS + 1;      // + the express list: the small tasks next to the chain, served by workgroups 1 .. DAG_EXPRESS_WGS only
constexpr int DAG_EXPRESS_WGS = 10;
constexpr int D_DARR = D_HEADS + 16 * N_LISTS;      // nt ints: arrivals of the split last panel at diagonal tile j
__host__ __device__ inline int tri(int i, int j) { return i * (i + 1) / 2 + j; }
__host__ __device__ inline int d_yprog(int nt) { return D_DARR + nt; }                           // i: panels applied to y_i
__host__ __device__ inline int d_stripc(int nt) { return d_yprog(nt) + nt; }                     // tri(i, k): strips of tile (i, k) done (8 = L_ik complete)
__host__ __device__ inline int d_usum(int nt) { return d_stripc(nt) + nt * (nt + 1) / 2; }       // tri(i, j): panels applied, summed over the tile's units
__host__ __device__ inline int d_uprog(int nt) { return d_usum(nt) + nt * (nt + 1) / 2; }        // 4 tri(i, j) + u: panels applied to unit u
inline int dag_state_ints(int nt) { return d_uprog(nt) + 4 * (nt * (nt + 1) / 2); }
// the group whose teams update tile (i, j): 4 x 2 blocks of tiles dealt round the eight groups (2.4 % apart in total work at 47 tile columns)
__host__ __device__ inline int tile_group(int i, int j) { return ((i >> 2) + 3 * (j >> 1)) & (N_GROUPS - 1); }

// ---- tasks: type | i | j | unit | k0 | nk in one 64-bit word
enum : unsigned { T_END = 0, T_STRIPS = 1, T_HALF = 2, T_QUARTER = 3, T_DIAG = 4, T_YSOLVE = 5, T_RHS = 6 };      // STRIPS: unit = first strip, nk = strips (8 or 4)
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
// units of tile (i, j): its 64 x 64 quarters q = (row half) + 2 (column half); the diagonal tile's upper-right quarter (q = 2) does not exist.
// A HALF task (unit = column half h) covers quarters 2 h and 2 h + 1: the form in which a tile absorbs panels while it lags behind the chain;
// the task that COMPLETES a tile goes in quarters (a quarter of the latency: the strips of the tile wait for it), and so does everything
// in the last columns (from quarter_from on), where tiles are too few to fill the teams otherwise.
__host__ __device__ inline int tile_units(int i, int j, int) { return i == j ? 3 : 4; }

constexpr int DAG_THREADS = 512;
constexpr int DAG_SPIN_LIMIT = 1 << 20;

typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(1))) int global_int;

struct DagArgs {
    double* S; double* y; double* x; double* Linv; double* Lpub; double* part; double* ok; double* stall;
    int* st;                               // state words
    const unsigned long long* tasks;       // the eight lists behind each other
    int list_off[N_LISTS], list_len[N_LISTS];
    int ld, nt, n_tasks, quarter_from, express_wgs;
    long long* trace;                      // development (tools/chol_test.hip built with -DDAG_TRACE): per task 4 stamps of the 100 MHz clock, behind them 2 per tile column of the chain
};

#ifdef DAG_TRACE
#define DAG_STAMP(slot) do { if (a.trace) a.trace[slot] = wall_clock64(); } while (0)
#define DAG_STAMP2(tr, slot) do { if (tr) (tr)[slot] = wall_clock64(); } while (0)
#else
#define DAG_STAMP2(tr, slot) do { } while (0)
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

// the same barrier where the wait is long (three wavefronts waiting for their leader's dependency polls): sleep between looks, so that
// they take neither LDS cycles nor issue slots from the other team's wavefronts on their SIMDs
__device__ __forceinline__ void team_sync_idle(Team& t)
{
    t.phase += 4;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (t.lane == 0) {
        __hip_atomic_fetch_add(t.cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(t.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < t.phase) __builtin_amdgcn_s_sleep(16);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// Panel solve of a tile by a team.  L_kk's 28 sub-diagonal 16 x 16 blocks (negated) and its 8 block inverses are parked in LDS in MFMA
// operand layout -- register r of lane l of block b at [b * 256 + r * 64 + l], every wave-wide read 512 contiguous bytes -- by the four
// wavefronts together (28 + 8 loads per lane, one trip); then every wavefront runs the strip of trsm_strip (chol_device.h) on its own
// 16-row strips, Y_c = Linv_cc (A_c^T - sum_{j<c} L_cj Y_j), same products in the same order: bit-identical.  No hand-off inside.
// (First form of this launch: one strip per task by four wavefronts handing Y_j round -- 8 648 tasks of 18 us, a third of all team time.)
// ---------------------------------------------------------------------------------------------
constexpr int LSH_DOUBLES = LPUB_BLOCKS * NB * NB, ISH_DOUBLES = NBLK * NB * NB;       // 7168 + 2048 doubles = 72 KB per team
// L_kk comes from where the chain workgroup publishes it: the scratch copy of its sub-diagonal blocks (block b = c (c - 1) / 2 + j
// column-major at Lpub_k + 256 b) and the row-major block inverses.  Items (block, chunk) = (item >> 2, item & 3) are dealt over the four
// wavefronts; inverse c is item 4 (LPUB_BLOCKS + c) + r.
__device__ __forceinline__ void park_Lkk(double* __restrict__ Lsh, double* __restrict__ Ish, const double* __restrict__ Lpub_k, const double* __restrict__ Linv_k, int tw, int ln)
{
    // all 36 loads of a lane first, then the LDS writes (item by item the compiler waited for every load before its write: 9.7 us per task)
    double v[LPUB_BLOCKS + NBLK];
#pragma unroll
    for (int q = 0; q < LPUB_BLOCKS + NBLK; ++q) {
        const int item = q * 4 + tw, b = item >> 2, r = item & 3;
        v[q] = b < LPUB_BLOCKS ? load_through(Lpub_k + b * 256 + (4 * r + (ln >> 4)) * NB + (ln & 15))
                               : load_through(Linv_k + (b - LPUB_BLOCKS) * NB * NB + (ln & 15) * NB + (ln >> 4) + 4 * r);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < LPUB_BLOCKS + NBLK; ++q) {
        const int item = q * 4 + tw, b = item >> 2, r = item & 3;
        if (b < LPUB_BLOCKS) Lsh[b * 256 + r * 64 + ln] = -v[q];
        else Ish[(b - LPUB_BLOCKS) * 256 + r * 64 + ln] = v[q];
    }
}
// one strip: element (strip row n = ln & 15, tile column col) at base[col * cstride]; `live` lanes hold real rows (the rhs strip has one)
__device__ __forceinline__ void strip_from_lds(double* __restrict__ base, size_t cstride, bool live, const double* __restrict__ Lsh, const double* __restrict__ Ish, int ln)
{
    double4_t Acc[NBLK], Y[NBLK];
#pragma unroll
    for (int c = 0; c < NBLK; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) Acc[c][r] = live ? base[(size_t)(c * NB + (ln >> 4) + 4 * r) * cstride] : 0.0;
#pragma unroll
    for (int c = 0; c < NBLK; ++c) {
        double4_t acc = Acc[c];
#pragma unroll
        for (int j = 0; j < c; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lsh[(c * (c - 1) / 2 + j) * 256 + r * 64 + ln], Y[j][r], acc, 0, 0, 0);
        double4_t yc = { 0, 0, 0, 0 };
#pragma unroll
        for (int r = 0; r < 4; ++r) yc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ish[c * 256 + r * 64 + ln], acc[r], yc, 0, 0, 0);
        Y[c] = yc;
        if (live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) store_through(base + (size_t)(c * NB + (ln >> 4) + 4 * r) * cstride, yc[r]);
        }
    }
}

// The strips of a tile NEXT TO THE CHAIN (one strip per wavefront), in three phases behind the in-tile factorisation of L_kk that runs
// at the same time on the chain workgroup:
//   block columns 0-3 in memory (progress word)  ->  steps 0-3 and the products of the later block columns with Y_0 .. Y_3  (26 of the 36 products)
//   block columns 4-5                             ->  steps 4, 5 and their products
//   tile factored (fact)                          ->  steps 6, 7: two inverse products and one update behind the last fetch
// Each phase parks what it needs in LDS first (26 / 7 / 3 loads per lane).  Per block column the products meet the accumulator in the
// order of the one-wavefront strip: bit-identical.  A team that starts when the tile is already factored does it in one go.
// Returns false when a wait ran out (the team abandons the task; the launch is being aborted).
__device__ __forceinline__ bool strips_phased(double* __restrict__ base, size_t cstride, const double* __restrict__ Lpub_k, const double* __restrict__ Linv_k,
                                              const int* __restrict__ st, int k, Team& t, double* __restrict__ Lsh, double* __restrict__ Ish, lds_int* ctl, long long* tr2, bool factored)
{
    const int ln = t.lane;
    const bool leader = t.tw == 0 && ln == 0;
    double4_t Acc[NBLK], Y[NBLK];
#pragma unroll
    for (int c = 0; c < NBLK; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) Acc[c][r] = base[(size_t)(c * NB + (ln >> 4) + 4 * r) * cstride];
    auto step = [&](int c) {          // Y_c = Linv_cc Acc_c, stored through
        double4_t yc = { 0, 0, 0, 0 };
#pragma unroll
        for (int r = 0; r < 4; ++r) yc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ish[c * 256 + r * 64 + ln], Acc[c][r], yc, 0, 0, 0);
        Y[c] = yc;
#pragma unroll
        for (int r = 0; r < 4; ++r) store_through(base + (size_t)(c * NB + (ln >> 4) + 4 * r) * cstride, yc[r]);
    };
    auto update = [&](int c, int j) { // Acc_c -= L(c, j) Y_j
#pragma unroll
        for (int r = 0; r < 4; ++r) Acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(Lsh[(c * (c - 1) / 2 + j) * 256 + r * 64 + ln], Y[j][r], Acc[c], 0, 0, 0);
    };
    auto wait_for = [&](const int* word, int target) {      // leader polls, everybody learns the outcome
        if (leader) ctl[5] = poll_ge(word, target, st + D_ABORT) ? 1 : 0;
        team_sync_idle(t);
        return __builtin_amdgcn_readfirstlane(ctl[5]) != 0;
    };
    // a phase parks blocks (c, j) with j in [j0, j1) and inverses [j0, i1): wavefront w takes chunk r = w of every block; a lane's loads all go
    // out before the first LDS write
    auto park_phase = [&](auto J0, auto J1, auto I1) {
        constexpr int j0 = decltype(J0)::value, j1 = decltype(J1)::value, i1 = decltype(I1)::value;
        const int r = t.tw;
        double v[LPUB_BLOCKS + NBLK];
#pragma unroll
        for (int c = 1; c < NBLK; ++c)
#pragma unroll
            for (int j = j0; j < j1; ++j)
                if (j < c) v[c * (c - 1) / 2 + j] = load_through(Lpub_k + (c * (c - 1) / 2 + j) * 256 + (4 * r + (ln >> 4)) * NB + (ln & 15));
#pragma unroll
        for (int c = j0; c < i1; ++c) v[LPUB_BLOCKS + c] = load_through(Linv_k + c * NB * NB + (ln & 15) * NB + (ln >> 4) + 4 * r);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int c = 1; c < NBLK; ++c)
#pragma unroll
            for (int j = j0; j < j1; ++j)
                if (j < c) Lsh[(c * (c - 1) / 2 + j) * 256 + r * 64 + ln] = -v[c * (c - 1) / 2 + j];
#pragma unroll
        for (int c = j0; c < i1; ++c) Ish[c * 256 + r * 64 + ln] = v[LPUB_BLOCKS + c];
    };
    // a team that arrives when the tile is already factored has nothing to overlap: one trip for all operands, then the plain strip
    if (!factored) {          // (a task that waited for `fact` at its dependency poll knows)
        if (leader) ctl[5] = ld_word(st + D_FACT) >= k + 1 ? 1 : 0;
        team_sync(t);
        factored = __builtin_amdgcn_readfirstlane(ctl[5]) != 0;
        team_sync(t);         // (ctl[5] is rewritten by the waits below: everybody has read it)
    }
    if (factored) {
        if (leader) DAG_STAMP2(tr2, 0);
        park_Lkk(Lsh, Ish, Lpub_k, Linv_k, t.tw, ln);
        team_sync(t);
        if (leader) DAG_STAMP2(tr2, 1);
#pragma unroll
        for (int c = 0; c < NBLK; ++c) {
#pragma unroll
            for (int j = 0; j < c; ++j) update(c, j);
            step(c);
        }
        if (leader) DAG_STAMP2(tr2, 2);
        return true;
    }
    // ---- phase 1: block columns 0 .. 3
    if (!wait_for(st + D_PROG, 8 * k + 3)) return false;
    park_phase(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
    team_sync(t);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int j = 0; j < c; ++j) update(c, j);
        step(c);
    }
#pragma unroll
    for (int c = 4; c < NBLK; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) update(c, j);
    // ---- phase 2: block columns 4, 5
    if (!wait_for(st + D_PROG, 8 * k + 5)) return false;
    park_phase(std::integral_constant<int, 4>{}, std::integral_constant<int, 6>{}, std::integral_constant<int, 6>{});
    team_sync(t);
    step(4);
    update(5, 4);
    step(5);
    update(6, 4); update(6, 5);
    update(7, 4); update(7, 5);
    // ---- phase 3: the tile is factored
    if (!wait_for(st + D_FACT, k + 1)) return false;
    park_phase(std::integral_constant<int, 6>{}, std::integral_constant<int, 7>{}, std::integral_constant<int, 8>{});
    team_sync(t);
    step(6);
    update(7, 6);
    step(7);
    return true;
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
        if (tid == 0) { *a.ok = 1.0; *a.stall = 0.0; }
        double* A = sm;                            // LayPacked: the 36 lower blocks
        double* Li = sm + PACKED_TILE_DOUBLES;     // 2 x (16 x 16)
        int* bail = reinterpret_cast<int*>(sm + PACKED_TILE_DOUBLES + 2 * NB * NB);
        if (tid == 0) *bail = 0;
        for (int k = 0; k < nt; ++k) {
            if (k > 0) {
                if (tid == 0 && (ld_word(st + D_INJECT) || !poll_ge(st + D_DARR + k, NDIAG, st + D_ABORT))) {
                    if (!ld_word(st + D_ABORT)) { *a.stall = 4.0; st_word(st + D_ABORT, 1); }
                    *bail = 1;
                }
                __syncthreads();
                if (*bail) return;
            }
            double* T = a.S + (size_t)(k * TILE) * ld + (size_t)k * TILE;
            if (tid == 0) DAG_STAMP(4 * (size_t)a.n_tasks + 2 * k);
            load_tile_packed_wt<DAG_THREADS>(A, T, ld, tid);
            __syncthreads();
            // block column by block column to the scratch copy + ONE progress word (potrf_tile_rows<.., 4>): the strips next to the chain work
            // in phases behind it; every strip reads L_kk from that copy and the block inverses, so the flag goes up as soon as those are in
            // memory and the factor itself goes to S afterwards, off the chain (the backward solve reads it there, a launch later)
            const bool failed = potrf_tile_rows<false, LayPacked, 4, DAG_THREADS / 64>(A, Li, a.Linv + (size_t)k * linv_stride, tid, NBLK,
                                                                     TilePublish{ a.Lpub + (size_t)k * LPUB_TILE_DOUBLES, st + D_PROG, 8 * k });
            if (tid == 0 && failed) *a.ok = 0.0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) { st_word(st + D_FACT, k + 1); DAG_STAMP(4 * (size_t)a.n_tasks + 2 * k + 1); }
            store_tile_packed_wt<DAG_THREADS>(T, A, ld, tid);
            if (k == 0) {
                // the chain opens the solve while it waits for tile 1: x pre-filled with the sentinel the backward substitution polls for,
                // and the backward solve's partial sums (same protocol)
                unsigned long long* xf = reinterpret_cast<unsigned long long*>(a.x);
                unsigned long long* pf = reinterpret_cast<unsigned long long*>(a.part);
                for (int i = tid; i < nt * TILE; i += DAG_THREADS) { xf[i] = X_SENTINEL; pf[i] = X_SENTINEL; }
            }
            __syncthreads();                 // (the tile's LDS is read by the store until here; the next tile overwrites it)
        }
        return;
    }

    // ================= worker teams =================
    // workgroups 1 .. DAG_EXPRESS_WGS serve the express list (the small tasks next to the chain must never queue behind a long update),
    // the others the list of their group
    const int team = wave >> 2, group = (int)blockIdx.x <= a.express_wgs ? N_GROUPS : (int)(blockIdx.x & (N_GROUPS - 1));
    Team t;
    double* const Lsh = sm + team * (LSH_DOUBLES + ISH_DOUBLES);                     // L_kk parked for the strips: sub-diagonal blocks, then inverses
    double* const Ish = Lsh + LSH_DOUBLES;
    lds_int* const ctl = (lds_int*)(sm + 2 * (LSH_DOUBLES + ISH_DOUBLES)) + team * 8;      // [0] barrier counter, [2..3] mailbox, [4] the task's index
    double* const yks = sm + 2 * (LSH_DOUBLES + ISH_DOUBLES) + 8 + team * TILE;             // y_k for a rhs task
    t.cnt = ctl; t.phase = 0; t.lane = lane; t.tw = wave & 3;
    if ((tid & 255) == 0) { ctl[0] = 0; ctl[2] = 0; ctl[3] = 0; ctl[4] = 0; ctl[5] = 0; }
    __syncthreads();                                           // the only workgroup-wide barrier: both teams are still together here
    const unsigned long long* const my_tasks = a.tasks + a.list_off[group];
    const int my_len = a.list_len[group];
    int* const head = st + D_HEADS + 16 * group;
    int* const yprog = st + d_yprog(nt);
    while (true) {
        // (the lane index is made opaque once per task: otherwise every address the strip and update shapes derive from it is hoisted out
        // of this loop, ~70 registers held across all roles and spilled)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        t.lane = ln;
        const bool leader = t.tw == 0 && ln == 0;
        if (leader) {
            unsigned long long w = 0;
            const int id = __hip_atomic_fetch_add((global_int*)head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (id < my_len && !ld_word(st + D_ABORT)) w = my_tasks[id];
            const size_t gid = (size_t)a.list_off[group] + id;
            (void)gid;
            if (w) DAG_STAMP(4 * gid);
            bool ready = true;
            const int i = t_i(w), j = t_j(w), u = t_unit(w), k0 = t_k0(w), nk = t_nk(w);
            switch (t_type(w)) {
                case T_STRIPS:     // tile (i, k = j) has absorbed its j panels; L_jj is in memory (k0 = 1: next to the chain -- the task follows the factorisation in phases instead)
                    ready = (k0 == 1 || poll_ge(st + D_FACT, j + 1, st + D_ABORT)) &&
                            (j == 0 || poll_ge(usum + tri(i, j), tile_units(i, j, a.quarter_from) * j, st + D_ABORT));
                    break;
                case T_YSOLVE:     // y_j has absorbed every panel before j; L_jj is in memory
                    ready = poll_ge(st + D_FACT, j + 1, st + D_ABORT) && poll_ge(yprog + j, j, st + D_ABORT);
                    break;
                case T_RHS:        // y_i stands at panel k0; L_i,k is complete and y_k solved for the last panel it takes (they imply the earlier ones)
                    ready = poll_ge(yprog + i, k0, st + D_ABORT) && poll_ge(stripc + tri(i, k0 + nk - 1), NBLK, st + D_ABORT) &&
                            poll_ge(st + D_YSOL, k0 + nk, st + D_ABORT);
                    break;
                case T_HALF:       // both quarters of the half stand at panel k0 (the diagonal tile's right half is quarter 3 alone) ...
                    ready = poll_ge(uprog + 4 * tri(i, j) + 2 * u + 1, k0, st + D_ABORT) && ((i == j && u == 1) || poll_ge(uprog + 4 * tri(i, j) + 2 * u, k0, st + D_ABORT)) &&
                            poll_ge(stripc + tri(i, k0 + nk - 1), NBLK, st + D_ABORT) && (i == j || poll_ge(stripc + tri(j, k0 + nk - 1), NBLK, st + D_ABORT));
                    break;
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
            if (w) DAG_STAMP(4 * gid + 1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // what the producers stored through is read plainly from here on
            if (w) DAG_STAMP(4 * gid + 2);
            ctl[2] = (int)(unsigned)w; ctl[3] = (int)(unsigned)(w >> 32); ctl[4] = (int)gid;
        }
        team_sync_idle(t);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane(ctl[2]), hi = (unsigned)__builtin_amdgcn_readfirstlane(ctl[3]);
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;
        const unsigned type = t_type(w);
        if (type == T_END) break;
        const int i = t_i(w), j = t_j(w), u = t_unit(w), k0 = t_k0(w), nk = t_nk(w);
        // the small tasks sit on somebody's critical path and share their SIMDs' matrix pipes with the other team's update: they go first
        if (type != T_HALF && type != T_QUARTER) __builtin_amdgcn_s_setprio(3);
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
        } else if (type == T_RHS) {
            // y_i -= L_ik y_k for k = k0 .. k0 + nk - 1: one chain of 128 products per row and panel (rows on wavefronts 0-1), y_k through LDS;
            // L_ik's row in four batches of 32 loads (left to the compiler the loads went out one by one inside the chain: 34 us per panel)
            const int r = (t.tw & 1) * 64 + ln;
            double* yi = a.y + (size_t)i * TILE + r;
            double yv = t.tw < 2 ? load_through(yi) : 0.0;
            for (int k = k0; k < k0 + nk; ++k) {
                if (t.tw < 2) yks[r] = load_through(a.y + (size_t)k * TILE + r);
                team_sync(t);
                if (t.tw < 2) {
                    const double* Lik = a.S + (size_t)(k * TILE) * ld + (size_t)i * TILE + r;
                    double accv = 0;
#pragma unroll
                    for (int c0 = 0; c0 < TILE; c0 += 32) {
                        double lv[32];
#pragma unroll
                        for (int c = 0; c < 32; ++c) lv[c] = Lik[(size_t)(c0 + c) * ld];
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int c = 0; c < 32; ++c) accv = __builtin_fma(lv[c], yks[c0 + c], accv);
                    }
                    yv -= accv;
                }
                team_sync(t);
            }
            if (t.tw < 2) store_through(yi, yv);
        } else {
            // T_STRIPS: strips u .. u + nk - 1 of tile (i, k = j);  T_YSOLVE: the rhs row y_j as a strip with one live row
            const double* Lpub_k = a.Lpub + (size_t)j * LPUB_TILE_DOUBLES;
            const double* Linv_k = a.Linv + (size_t)j * linv_stride;
            if (type == T_STRIPS && nk == 4) {
                if (!strips_phased(a.S + (size_t)(j * TILE) * ld + (size_t)i * TILE + (u + t.tw) * NB + (ln & 15), (size_t)ld, Lpub_k, Linv_k, st, j, t, Lsh, Ish, ctl, a.trace ? a.trace + 4 * (size_t)a.n_tasks + 2 * nt + 16 + 4 * (size_t)ctl[4] : nullptr, k0 == 0)) {
                    if (leader && !ld_word(st + D_ABORT)) { *a.stall = 4.0; st_word(st + D_ABORT, 1); }
                }
            } else {
                park_Lkk(Lsh, Ish, Lpub_k, Linv_k, t.tw, ln);
                team_sync(t);
                if (type == T_YSOLVE) {
                    if (t.tw == 0) strip_from_lds(a.y + (size_t)j * TILE, 1, (ln & 15) == 0, Lsh, Ish, ln);
                } else {
                    for (int sidx = u + t.tw; sidx < u + nk; sidx += 4)
                        strip_from_lds(a.S + (size_t)(j * TILE) * ld + (size_t)i * TILE + sidx * NB + (ln & 15), (size_t)ld, true, Lsh, Ish, ln);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wavefront's stores are in memory
        __builtin_amdgcn_s_setprio(0);
        team_sync(t);
        if (leader) {
            DAG_STAMP(4 * (size_t)ctl[4] + 3);
            switch (type) {
                case T_STRIPS: add_word(stripc + tri(i, j), nk); break;
                case T_YSOLVE: st_word(st + D_YSOL, j + 1); break;
                case T_RHS: st_word(yprog + i, k0 + nk); break;
                case T_DIAG: add_word(st + D_DARR + j, 1); break;
                case T_HALF:
                    st_word(uprog + 4 * tri(i, j) + 2 * u + 1, k0 + nk);
                    if (!(i == j && u == 1)) st_word(uprog + 4 * tri(i, j) + 2 * u, k0 + nk);
                    add_word(usum + tri(i, j), (i == j && u == 1) ? nk : 2 * nk);
                    break;
                default: st_word(uprog + 4 * tri(i, j) + u, k0 + nk); add_word(usum + tri(i, j), nk); break;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The schedule: list scheduling of the task graph in simulated time (costs in us from the stamps of tools/_bin/chol_test_trace).
// ---------------------------------------------------------------------------------------------
struct SimCosts {
    // (round 5's measured costs, kept as the ORDER they produce: with the round-6 tile factorisation -- potrf 17, early 7 -- the launch
    // measured the same, 1.967 against 1.972 ms at 47 tile columns, and one random skyline of tests/test_chol_schedule.py gave lists that
    // check_schedule rejects, i.e. a launch that would fall back to column-by-column)
    double potrf = 21.5;                  // tile -> LDS, in-tile factorisation, publication drained, flag
    double early = 9.0;                   // into the tile: block columns 0-3 are published (the phased strips' first phase may start)
    double strips4 = 12.0, tail = 4.5;    // four strips of a tile by one team (a wavefront each); phased: ends no earlier than `tail` after the tile is factored
    double ysolve = 12.0;
    double rhs0 = 3.0, rhs1 = 6.0;        // rhs rows: fixed + per panel
    double diag = 5.0, gather = 1.5;      // one ninth of the split panel; ninth arrival -> chain workgroup sees it
    double hop = 1.5;                     // a predecessor on another compute unit becomes visible to a poll
    double half0 = 4.0, half1 = 15.2;     // half-tile task: fixed (C in and out, ramp, drain, hand-off) + per panel
    double quarter0 = 3.3, quarter1 = 8.5;
};

struct Ev {
    double t; long seq; int kind, a, b, c, d;
    bool operator>(const Ev& o) const { return t != o.t ? t > o.t : seq > o.seq; }
};
enum { EV_SREADY, EV_SDONE, EV_DREADY, EV_DDONE, EV_UDONE, EV_POTRF, EV_YREADY, EV_YDONE, EV_UREADY, EV_RDONE, EV_PEARLY };

struct Schedule {
    std::vector<unsigned long long> lists[N_LISTS];
    int quarter_from = 0, express_wgs = DAG_EXPRESS_WGS;
};
// Workgroups on the express list.  Dense systems: ten (8 ... 24 measured the same, round 6).  With an envelope nearly every task sits next to
// the chain and a team that holds a task it waits for is what makes a hand-off one poll: 10 / 16 / 32 / 48 / 64 workgroups -> 1.84 / 1.65 /
// 1.63 / 1.59 / 1.57 ms at 47 tile columns (twenty teams were all holding tasks all the time, the list's later entries waited for a free one)
inline int express_wgs_for(bool envelope, int n_cu)
{
    static const int env_wgs = [] { const char* e = std::getenv("MAGE_CHOL_DAG_ENV_EXPRESS_WGS"); return e ? std::max(4, std::min(64, std::atoi(e))) : 48; }();
    static const int dense_wgs = [] { const char* e = std::getenv("MAGE_CHOL_DAG_EXPRESS_WGS"); return e ? std::max(4, std::min(64, std::atoi(e))) : DAG_EXPRESS_WGS; }();
    return std::min(envelope ? env_wgs : dense_wgs, std::max(4, (n_cu - 1 - 2 * N_GROUPS) / 4));
}

// teams of group g when n_cu workgroups are launched (workgroup b belongs to group b % 8; workgroup 0 is the chain)
inline int group_teams(int g, int n_cu, int express_wgs)
{
    if (g == N_GROUPS) return 2 * express_wgs;
    int n = 0;
    for (int b = express_wgs + 1; b < n_cu; ++b) n += (b & (N_GROUPS - 1)) == g;
    return 2 * n;
}

// `env` (optional, nt entries): env[i] = the first tile column of tile row i that can hold a non-zero (the skyline of S; env[i] <= i).
// Tiles left of it are zero and stay zero in the factor, so: no strips for them, tile (i, j) starts at panel max(env[i], env[j]) instead
// of 0, y_i at panel env[i].  The launch's state then starts from an IMAGE (envelope_state) in which those panels count as applied; the
// kernel and the checks are unchanged.  A dense system is env = all zeros.
Schedule build_schedule(int nt, int n_cu, int gmax, const int* env = nullptr)
{
    const SimCosts C;
    Schedule out;
    auto env_of = [&](int i) { return env ? env[i] : 0; };
    auto kstart = [&](int i, int j) { return std::max(env_of(i), env_of(j)); };
    int n_teams = 0, free_teams[N_LISTS];
    // (with an envelope nearly every task sits next to the chain: more workgroups on the express list)
    const int express_wgs = express_wgs_for(env != nullptr, n_cu);
    out.express_wgs = express_wgs;
    for (int g = 0; g < N_LISTS; ++g) { free_teams[g] = group_teams(g, n_cu, express_wgs); if (g < N_GROUPS) n_teams += free_teams[g]; }
    // quarter tiles from the first column whose trailing matrix no longer offers a half-tile task per team
    int& quarter_from = out.quarter_from;
    quarter_from = nt;
    for (int j = 1; j < nt; ++j) { const int m = nt - j; if (m * (m + 1) < n_teams) { quarter_from = j; break; } }
    if (env) quarter_from = 1;          // a band offers a few tiles per column: quarters from the start
    const int ntri = nt * (nt + 1) / 2;
    std::vector<int> strip_cnt(ntri, 0), availc(ntri, 0), nxt(4 * ntri, 0), darr(nt, 0), yprog(nt, 0);
    std::vector<char> busy(4 * ntri, 0), queued(4 * ntri, 0), tile_done(ntri, 0), strips_out(ntri, 0), diag_out(nt, 0), y_out(nt, 0), rhs_busy(nt, 0);
    std::vector<double> potrf_end(nt, 0.0);
    std::vector<char> pearly(nt, 0);
    int fact = 0, ysol = 0;
    std::priority_queue<Ev, std::vector<Ev>, std::greater<Ev>> events;
    long seq = 0;
    auto post = [&](double t, int kind, int a2 = 0, int b = 0, int c = 0, int d = 0) { events.push(Ev{ t, seq++, kind, a2, b, c, d }); };
    typedef std::tuple<int, int, long, int, int, int> Key;      // (column, row, seq, i, j, unit)
    typedef std::priority_queue<Key, std::vector<Key>, std::greater<Key>> KeyQueue;
    std::vector<Key> ready_units[N_LISTS];
    KeyQueue ready_strips, ready_near_strips;                  // updates wait for a team of their tile's group (next to the diagonal: for an express team); the strips of far tiles go where a team is free
    std::vector<std::pair<int, int>> ready_diag;
    std::vector<int> ready_y;
    double now = 0;
    auto has_quarter = [&](int i, int j, int q) { return !(i == j && q == 2); };
    auto limit_of = [&](int i, int j) { return i == j ? j - 1 : j; };       // panels the regular units apply (the diagonal tile's last one is split)
    auto tile_complete = [&](int i, int j) {
        for (int q = 0; q < 4; ++q) if (has_quarter(i, j, q) && nxt[4 * tri(i, j) + q] < limit_of(i, j)) return false;
        return true;
    };
    // the rows next to the chain in two halves, a wavefront per strip, PHASED against the factorisation of L_kk (they may start once its
    // first block columns are published); the others as one task once the tile is factored
    auto release_strips = [&](int i, int k, double t) {
        if (strips_out[tri(i, k)] || !(k == 0 || tile_done[tri(i, k)])) return;
        const bool near = i <= k + 2;
        if (near ? !pearly[k] : fact < k + 1) return;
        strips_out[tri(i, k)] = 1;
        post(t, EV_SREADY, i, k, 0, near ? 12 : 4); post(t, EV_SREADY, i, k, 4, near ? 12 : 4);       // two tasks of four strips, a wavefront per strip (d = strips | 8 for the phased ones)
    };
    auto consider = [&](int i, int j, double t) {
        if (j < env_of(i)) return;          // (a tile left of its row's envelope: nothing to do, ever)
        int& av = availc[tri(i, j)];
        while (av < j && strip_cnt[tri(i, av)] == NBLK && strip_cnt[tri(j, av)] == NBLK) ++av;
        const int lim = std::min(av, limit_of(i, j));
        if (j >= quarter_from) {
            for (int q = 0; q < 4; ++q) {
                const int id = 4 * tri(i, j) + q;
                if (has_quarter(i, j, q) && !busy[id] && !queued[id] && nxt[id] < lim) { queued[id] = 1; post(t, EV_UREADY, i, j, q); }
            }
        } else {
            for (int h = 0; h < 2; ++h) {       // the half is queued as one entry (unit code 4 + h) while its quarters stand together and idle
                bool idle = true;
                for (int q = 2 * h; q < 2 * h + 2; ++q) if (has_quarter(i, j, q)) { const int id = 4 * tri(i, j) + q; idle = idle && !busy[id] && !queued[id] && nxt[id] < lim; }
                if (!idle) continue;
                for (int q = 2 * h; q < 2 * h + 2; ++q) if (has_quarter(i, j, q)) queued[4 * tri(i, j) + q] = 1;
                post(t, EV_UREADY, i, j, 4 + h);
            }
        }
        if (i == j && !diag_out[j] && av >= j && tile_complete(j, j)) {
            diag_out[j] = 1;
            for (int p = 0; p < NDIAG; ++p) post(t, EV_DREADY, j, p);
        }
    };
    auto start_potrf = [&](int k, double t) {
        const double t0 = std::max(t, k > 0 ? potrf_end[k - 1] : 0.0);
        potrf_end[k] = t0 + C.potrf;
        post(t0 + C.early, EV_PEARLY, k);
        post(potrf_end[k], EV_POTRF, k);
    };
    auto any_free = [&] { int g = 0; for (int x = 1; x < N_GROUPS; ++x) if (free_teams[x] > free_teams[g]) g = x; return free_teams[g] > 0 ? g : -1; };
    auto express_free = [&] { return free_teams[N_GROUPS] > 0 ? N_GROUPS : -1; };
    auto near_diag = [&](int i, int j) { return i - j <= 2; };      // the COMPLETING quarters of these tiles are express tasks (their strips / split panel wait for them)
    // rhs rows: y_i may absorb panels yprog[i] .. min(ysol, strips of row i complete) - 1
    auto rhs_avail = [&](int i) {
        int k = yprog[i];
        while (k < i && k < ysol && strip_cnt[tri(i, k)] == NBLK) ++k;
        return k - yprog[i];
    };
    if (env) {
        // what the envelope makes moot counts as done from the start
        for (int i = 1; i < nt; ++i) {
            yprog[i] = std::min(env_of(i), i);
            for (int k = 0; k < i; ++k) {
                if (k < env_of(i)) { strip_cnt[tri(i, k)] = NBLK; strips_out[tri(i, k)] = 1; tile_done[tri(i, k)] = 1; }
            }
            for (int j = 1; j <= i; ++j) {
                const int first = j < env_of(i) ? limit_of(i, j) : std::min(kstart(i, j), limit_of(i, j));
                for (int q = 0; q < 4; ++q) nxt[4 * tri(i, j) + q] = std::max(first, 0);
                if (i != j && tile_complete(i, j)) tile_done[tri(i, j)] = 1;
            }
            if (env_of(i) >= i) { darr[i] = NDIAG; diag_out[i] = 1; }          // a diagonal tile nothing left of it touches: no split panel
        }
    }
    start_potrf(0, 0.0);
    while (true) {
        bool progress = true;
        while (progress) {
            progress = false;
            int g;
            while (!ready_diag.empty() && (g = express_free()) >= 0) {
                const auto [j, p] = ready_diag.back(); ready_diag.pop_back();
                out.lists[g].push_back(task_word(T_DIAG, j, j, p, j - 1, 1));
                --free_teams[g]; post(now + C.diag, EV_DDONE, j, p, g); progress = true;
            }
            while (!ready_y.empty() && (g = express_free()) >= 0) {
                const int k = ready_y.back(); ready_y.pop_back();
                out.lists[g].push_back(task_word(T_YSOLVE, k, k, 0, 0, 0));
                --free_teams[g]; post(now + C.ysolve, EV_YDONE, k, g); progress = true;
            }
            while (!ready_near_strips.empty() && (g = express_free()) >= 0) {
                const Key key = ready_near_strips.top(); ready_near_strips.pop();
                const int i = std::get<3>(key), k = std::get<4>(key), s0 = std::get<5>(key) & 15;
                out.lists[g].push_back(task_word(T_STRIPS, i, k, s0, 1, 4));
                --free_teams[g]; post(std::max(now + C.strips4, potrf_end[k] + C.tail), EV_SDONE, i, k, 4, g); progress = true;
            }
            while (!ready_strips.empty() && (g = any_free()) >= 0) {
                const Key key = ready_strips.top(); ready_strips.pop();
                const int i = std::get<3>(key), k = std::get<4>(key), s0 = std::get<5>(key) & 15, phased = (std::get<5>(key) >> 7) & 1;
                out.lists[g].push_back(task_word(T_STRIPS, i, k, s0, phased, 4));
                --free_teams[g]; post(phased ? std::max(now + C.strips4, potrf_end[k] + C.tail) : now + C.strips4, EV_SDONE, i, k, 4, g); progress = true;
            }
            // rhs rows, lowest row first (the row the next solve waits for)
            for (int i = 1; i < nt; ++i) {
                if (rhs_busy[i]) continue;
                if ((g = (i <= ysol + 1 ? express_free() : any_free())) < 0) { if (i > ysol + 1) break; else continue; }
                const int n = std::min(rhs_avail(i), gmax);
                if (n <= 0) continue;
                if (yprog[i] + n < i && n < gmax && i > ysol + 1) continue;      // a far row waits until it can take a full task (or everything it still needs)
                out.lists[g].push_back(task_word(T_RHS, i, i, 0, yprog[i], n));
                rhs_busy[i] = 1; --free_teams[g]; post(now + C.rhs0 + n * C.rhs1, EV_RDONE, i, n, g); progress = true;
            }
            for (g = 0; g < N_LISTS; ++g) {
                while (free_teams[g] > 0 && !ready_units[g].empty()) {
                    // Which unit next: the ones the chain will need soon first (earliest column), the others by how many panels they can take in
                    // one task (a block of C is read and written once per task: a far tile served eagerly costs a task per panel)
                    size_t best = 0;
                    std::tuple<int, int, int, int, long> best_key{ 2, 0, 0, 0, 0 };
                    for (size_t x = 0; x < ready_units[g].size(); ++x) {
                        const Key& c = ready_units[g][x];
                        const int ci = std::get<3>(c), cj = std::get<4>(c), cu = std::get<5>(c);
                        const int cid = 4 * tri(ci, cj) + (cu >= 4 ? 2 * (cu - 4) + 1 : cu);
                        const int avail = std::min(std::min(availc[tri(ci, cj)], limit_of(ci, cj)) - nxt[cid], gmax);
                        const bool urgent = g == N_GROUPS || cj - fact <= 2 + avail / 2 || avail >= gmax;
                        const std::tuple<int, int, int, int, long> k2 = urgent ? std::make_tuple(0, cj, ci, 0, std::get<2>(c)) : std::make_tuple(1, -avail, cj, ci, std::get<2>(c));
                        if (k2 < best_key) { best_key = k2; best = x; }
                    }
                    const Key key = ready_units[g][best];
                    ready_units[g][best] = ready_units[g].back(); ready_units[g].pop_back();
                    const int i = std::get<3>(key), j = std::get<4>(key), u = std::get<5>(key);
                    const int lim = std::min(availc[tri(i, j)], limit_of(i, j));
                    if (u >= 4) {
                        // a half: quarters 2 h, 2 h + 1 (they stand at the same panel)
                        const int h = u - 4, q1 = 2 * h + 1, q0 = has_quarter(i, j, 2 * h) ? 2 * h : q1, id1 = 4 * tri(i, j) + q1;
                        const int n = std::min(lim - nxt[id1], gmax);
                        if (n <= 0) { queued[id1] = 0; queued[4 * tri(i, j) + q0] = 0; continue; }
                        if (nxt[id1] + n == limit_of(i, j)) {
                            // this task completes the tile: in quarters (they stay queued and come up again right away, one team each)
                            const int dst = near_diag(i, j) ? N_GROUPS : g;
                            ready_units[dst].push_back(Key{ j, i, seq++, i, j, q0 });
                            if (q1 != q0) ready_units[dst].push_back(Key{ j, i, seq++, i, j, q1 });
                            continue;
                        }
                        queued[id1] = 0; queued[4 * tri(i, j) + q0] = 0;
                        out.lists[g].push_back(task_word(T_HALF, i, j, h, nxt[id1], n));
                        busy[id1] = 1; busy[4 * tri(i, j) + q0] = 1; --free_teams[g];
                        post(now + C.half0 + n * C.half1, EV_UDONE, i, j, u, n | (g << 8));
                        progress = true;
                        continue;
                    }
                    const int id = 4 * tri(i, j) + u;
                    const int n = std::min(lim - nxt[id], gmax);
                    if (n <= 0) { queued[id] = 0; continue; }
                    if (g != N_GROUPS && near_diag(i, j) && nxt[id] + n == limit_of(i, j)) { ready_units[N_GROUPS].push_back(Key{ j, i, seq++, i, j, u }); continue; }      // (stays queued)
                    queued[id] = 0;
                    out.lists[g].push_back(task_word(T_QUARTER, i, j, u, nxt[id], n));
                    busy[id] = 1; --free_teams[g];
                    post(now + C.quarter0 + n * C.quarter1, EV_UDONE, i, j, u, n | (g << 8));
                    progress = true;
                }
            }
        }
        if (events.empty()) break;
        now = events.top().t;
        while (!events.empty() && events.top().t <= now) {      // everything that happens at this instant, then dispatch
            const Ev e = events.top(); events.pop();
            switch (e.kind) {
                case EV_POTRF: {
                    const int k = e.a;
                    fact = k + 1;
                    if (k + 1 < nt && env && env_of(k + 1) >= k + 1) start_potrf(k + 1, now + C.gather);
                    if (!y_out[k] && yprog[k] >= k) { y_out[k] = 1; post(now + C.hop, EV_YREADY, k); }
                    for (int i = k + 1; i < nt; ++i) release_strips(i, k, now + C.hop);
                    break;
                }
                case EV_PEARLY:
                    pearly[e.a] = 1;
                    for (int i = e.a + 1; i < nt && i <= e.a + 2; ++i) release_strips(i, e.a, now);
                    break;
                case EV_SREADY: ((e.d & 8) ? ready_near_strips : ready_strips).push(Key{ e.b, e.a, seq++, e.a, e.b, e.c | (e.d << 4) }); break;
                case EV_YREADY: ready_y.push_back(e.a); break;
                case EV_DREADY: ready_diag.push_back({ e.a, e.b }); break;
                case EV_UREADY: ready_units[tile_group(e.a, e.b)].push_back(Key{ e.b, e.a, seq++, e.a, e.b, e.c }); break;
                case EV_YDONE: ++free_teams[e.b]; ysol = e.a + 1; break;
                case EV_RDONE: {
                    ++free_teams[e.c];
                    const int i = e.a;
                    rhs_busy[i] = 0; yprog[i] += e.b;
                    if (yprog[i] >= i && fact >= i + 1 && !y_out[i]) { y_out[i] = 1; post(now + C.hop, EV_YREADY, i); }
                    break;
                }
                case EV_DDONE:
                    ++free_teams[e.c];
                    if (++darr[e.a] == NDIAG) start_potrf(e.a, now + C.hop + C.gather);
                    break;
                case EV_SDONE: {
                    ++free_teams[e.d];
                    const int i = e.a, k = e.b;
                    strip_cnt[tri(i, k)] += e.c;
                    if (strip_cnt[tri(i, k)] == NBLK) {
                        for (int j = k + 1; j <= i; ++j) consider(i, j, now + C.hop);
                        for (int i2 = i + 1; i2 < nt; ++i2) consider(i2, i, now + C.hop);
                    }
                    break;
                }
                case EV_UDONE: {
                    const int i = e.a, j = e.b, u = e.c;
                    const int n = e.d & 0xff;
                    ++free_teams[e.d >> 8];
                    if (u >= 4) { for (int q = 2 * (u - 4); q < 2 * (u - 4) + 2; ++q) if (has_quarter(i, j, q)) { busy[4 * tri(i, j) + q] = 0; nxt[4 * tri(i, j) + q] += n; } }
                    else { busy[4 * tri(i, j) + u] = 0; nxt[4 * tri(i, j) + u] += n; }
                    if (i != j && !tile_done[tri(i, j)] && tile_complete(i, j)) {
                        tile_done[tri(i, j)] = 1;
                        release_strips(i, j, now + C.hop);
                    }
                    consider(i, j, now);
                    break;
                }
                default: break;
            }
        }
    }
    return out;
}

// The lists cannot deadlock teams that take them in order: replay them with instantaneous tasks -- every group advances its cursor while
// the task under it has its wait conditions met (the conditions the kernel polls) -- and require that all lists run out; the chain
// advances when the nine arrivals of its next tile are in.  Also checks completeness: every strip, every panel of every unit, every split
// panel, every rhs panel and solve exactly once and in range.
// The progress words as a launch finds them: zero for a dense system; with an envelope (build_schedule) every panel that multiplies a
// structurally zero tile counts as applied.  One rule for the checker below and for the image the launch copies over its state words.
void initial_progress(int nt, const int* env, int* stripc, int* usum, int* uprog, int* darr, int* yprog)
{
    if (!env) return;
    for (int i = 1; i < nt; ++i) {
        yprog[i] = std::min(env[i], i);
        for (int k = 0; k < i && k < env[i]; ++k) stripc[tri(i, k)] = NBLK;
        for (int j = std::max(env[i], 1); j <= i; ++j) {
            const int limit = i == j ? j - 1 : j, first = std::max(0, std::min(std::max(env[i], env[j]), limit)), units = tile_units(i, j, 0);
            for (int q = 0; q < 4; ++q) uprog[4 * tri(i, j) + q] = first;
            usum[tri(i, j)] = units * first;
        }
        if (env[i] >= i) darr[i] = NDIAG;
    }
}

bool check_schedule(const Schedule& sch, int nt, const int* env = nullptr)
{
    const int qf = sch.quarter_from, ntri = nt * (nt + 1) / 2;
    std::vector<int> stripc(ntri, 0), usum(ntri, 0), uprog(4 * ntri, 0), darr(nt, 0), yprog(nt, 0);
    initial_progress(nt, env, stripc.data(), usum.data(), uprog.data(), darr.data(), yprog.data());
    int fact = 1, ysol = 0;          // potrf(0) depends on nothing
    while (fact < nt && darr[fact] == NDIAG) ++fact;          // (... nor does a diagonal tile the envelope cuts off from everything left of it)
    size_t cur[N_LISTS] = {};
    bool moved = true;
    while (moved) {
        moved = false;
        for (int g = 0; g < N_LISTS; ++g) {
            while (cur[g] < sch.lists[g].size()) {
                const unsigned long long w = sch.lists[g][cur[g]];
                const int i = t_i(w), j = t_j(w), u = t_unit(w), k0 = t_k0(w), nk = t_nk(w);
                bool ok = false;
                switch (t_type(w)) {
                    case T_STRIPS:
                        if (i <= j || nk != 4 || (u != 0 && u != 4) || k0 > 1) return false;
                        ok = fact >= j + 1 && (j == 0 || usum[tri(i, j)] >= tile_units(i, j, qf) * j);
                        if (ok) stripc[tri(i, j)] += nk;
                        break;
                    case T_YSOLVE:
                        ok = fact >= j + 1 && yprog[j] >= j;
                        if (ok) { if (ysol != j) return false; ysol = j + 1; }
                        break;
                    case T_RHS:
                        if (nk < 1 || k0 + nk > i) return false;
                        ok = yprog[i] >= k0 && stripc[tri(i, k0 + nk - 1)] >= NBLK && ysol >= k0 + nk;
                        if (ok) { if (yprog[i] != k0) return false; yprog[i] = k0 + nk; }
                        break;
                    case T_HALF: {
                        if (nk < 1 || k0 + nk > (i == j ? j - 1 : j) || tile_group(i, j) != g || j >= qf || u > 1) return false;
                        const int q1 = 4 * tri(i, j) + 2 * u + 1, q0 = (i == j && u == 1) ? q1 : q1 - 1;
                        ok = uprog[q1] >= k0 && uprog[q0] >= k0 && stripc[tri(i, k0 + nk - 1)] >= NBLK && (i == j || stripc[tri(j, k0 + nk - 1)] >= NBLK);
                        if (ok) {
                            if (uprog[q1] != k0 || uprog[q0] != k0) return false;
                            uprog[q1] = k0 + nk; uprog[q0] = k0 + nk; usum[tri(i, j)] += q0 == q1 ? nk : 2 * nk;
                        }
                        break;
                    }
                    case T_QUARTER:
                        if (nk < 1 || k0 + nk > (i == j ? j - 1 : j) || u > 3 || (i == j && u == 2)) return false;
                        if (g != tile_group(i, j) && !(g == N_GROUPS && i - j <= 2 && k0 + nk == (i == j ? j - 1 : j))) return false;
                        ok = uprog[4 * tri(i, j) + u] >= k0 && stripc[tri(i, k0 + nk - 1)] >= NBLK && (i == j || stripc[tri(j, k0 + nk - 1)] >= NBLK);
                        if (ok) { if (uprog[4 * tri(i, j) + u] != k0) return false; uprog[4 * tri(i, j) + u] = k0 + nk; usum[tri(i, j)] += nk; }
                        break;
                    case T_DIAG:
                        ok = stripc[tri(j, j - 1)] >= NBLK && usum[tri(j, j)] >= tile_units(j, j, qf) * (j - 1);
                        if (ok) { ++darr[j]; while (fact < nt && darr[fact] == NDIAG) ++fact; }
                        break;
                    default: return false;
                }
                if (!ok) break;
                ++cur[g]; moved = true;
            }
        }
    }
    for (int g = 0; g < N_LISTS; ++g) if (cur[g] != sch.lists[g].size()) return false;
    if (fact != nt || ysol != nt) return false;
    for (int i = 1; i < nt; ++i) {
        if (yprog[i] != i) return false;
        for (int j = 0; j < i; ++j) if (stripc[tri(i, j)] != NBLK) return false;
        for (int j = env ? std::max(env[i], 1) : 1; j <= i; ++j) if (usum[tri(i, j)] != tile_units(i, j, qf) * (i == j ? j - 1 : j)) return false;
    }
    return true;
}

struct DagSchedule {
    unsigned long long* d_tasks = nullptr;
    int* d_image = nullptr;          // with an envelope: the state words a launch starts from (copied over them instead of the zero-fill)
    int* d_kmax = nullptr;           // ... and per tile column the last tile row that can hold a non-zero (the backward solve stops there)
    int n_tasks = 0, quarter_from = 0, express_wgs = DAG_EXPRESS_WGS, n_cu = 0, off[N_LISTS] = {}, len[N_LISTS] = {};
    bool ok = false;
    unsigned long long last_use = 0;
};
typedef std::tuple<int, int, unsigned long long> SchedKey;          // (device, tile columns, hash of the envelope: 0 = dense)
unsigned long long env_hash(const int* env, int nt)
{
    if (!env) return 0;
    unsigned long long h = 1469598103934665603ull;
    for (int i = 0; i < nt; ++i) h = (h ^ (unsigned long long)(unsigned)env[i]) * 1099511628211ull;
    return h | 1ull;
}
// The lists of a system size are built ONCE per (device, tile count) -- and not by the thread that asks for them: the simulation takes
// 19 ms at 47 tile columns and 140 ms at 94, which the first factorisation after a loop closure (a new map size:
// Tasks/LoopClosureWorker.cpp:163-208 in the reference) would pay before its first launch.  A worker thread builds and checks them;
// until they are there chol_dag_factor says "not this time" and the factorisation goes column by column (the same bits, 0.7 ms slower
// at 47 tile columns).  chol_dag_prefetch starts the job as soon as the system's order is known (the structure build), i.e. a few
// milliseconds before the first factorisation asks.
struct SchedJob {
    std::mutex m;
    std::condition_variable cv;
    bool done = false, valid = false;
    int quarter_from = 0, express_wgs = DAG_EXPRESS_WGS, n_cu = 0, off[N_LISTS] = {}, len[N_LISTS] = {};
    std::vector<unsigned long long> flat;
    std::vector<int> env, image, kmax;          // (empty: dense)
    double build_ms = 0;
};
std::mutex g_sched_mutex;
std::map<SchedKey, DagSchedule> g_sched;                  // uploaded
std::map<SchedKey, std::shared_ptr<SchedJob>> g_jobs;    // being built (or built and not yet uploaded)
std::map<int, int> g_dev_cu;                             // compute units per device (chol_dag_init_device), under g_sched_mutex
unsigned long long g_use_clock = 0;
std::atomic<double> g_last_build_ms{ 0.0 };
constexpr size_t SCHED_CACHE_MAX = 16;                   // uploaded lists kept per process (least recently used goes first; ~170 KB each at 47 tile columns)
struct Turnstile { hipEvent_t ev = nullptr; hipStream_t last = nullptr; bool recorded = false, several = false; };
std::mutex g_turn_mutex;
std::map<int, Turnstile> g_turn;                         // per device: the event behind the latest task-graph launch and the stream it went to
std::atomic<int> g_inject_stalls{ 0 };      // mage_debug_chol_inject_stall: that many launches from now on behave as if a wait had run out

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

int device_cus(int dev)          // (g_sched_mutex held)
{
    auto it = g_dev_cu.find(dev);
    return it == g_dev_cu.end() ? 0 : it->second;
}

// (g_sched_mutex held) the job that builds the lists of a key, started if there is none
std::shared_ptr<SchedJob> job_for(const SchedKey& key, int nt, int n_cu, const int* env)
{
    auto it = g_jobs.find(key);
    if (it != g_jobs.end()) return it->second;
    auto job = std::make_shared<SchedJob>();
    job->n_cu = n_cu;
    if (env) job->env.assign(env, env + nt);
    g_jobs[key] = job;
    const int gmax = dag_fuse_max();
    std::thread([job, nt, n_cu, gmax] {
        const auto t0 = std::chrono::steady_clock::now();
        const int* e = job->env.empty() ? nullptr : job->env.data();
        const Schedule sch = build_schedule(nt, n_cu, gmax, e);
        std::vector<unsigned long long> flat;
        int off[N_LISTS], len[N_LISTS];
        for (int g = 0; g < N_LISTS; ++g) { off[g] = (int)flat.size(); len[g] = (int)sch.lists[g].size(); flat.insert(flat.end(), sch.lists[g].begin(), sch.lists[g].end()); }
        const bool valid = check_schedule(sch, nt, e);
        std::vector<int> image, kmax;
        if (e) {
            image.assign((size_t)dag_state_ints(nt), 0);
            initial_progress(nt, e, image.data() + d_stripc(nt), image.data() + d_usum(nt), image.data() + d_uprog(nt), image.data() + D_DARR, image.data() + d_yprog(nt));
            kmax.assign(nt, 0);
            for (int j = 0; j < nt; ++j) { kmax[j] = j; for (int i = j + 1; i < nt; ++i) if (e[i] <= j) kmax[j] = i; }
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::lock_guard<std::mutex> lock(job->m);
        job->flat.swap(flat); job->image.swap(image); job->kmax.swap(kmax); job->quarter_from = sch.quarter_from; job->express_wgs = sch.express_wgs; job->valid = valid; job->build_ms = ms;
        for (int g = 0; g < N_LISTS; ++g) { job->off[g] = off[g]; job->len[g] = len[g]; }
        job->done = true;
        job->cv.notify_all();
    }).detach();
    return job;
}

// The uploaded lists of (current device, nt); nullptr while they are being built (wait = false) or when they cannot be had.
const DagSchedule* get_schedule(int nt, bool wait, const int* env = nullptr)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    const SchedKey key{ dev, nt, env_hash(env, nt) };
    std::unique_lock<std::mutex> lock(g_sched_mutex);
    auto it = g_sched.find(key);
    if (it == g_sched.end()) {
        const int n_cu = device_cus(dev);
        if (n_cu <= 0) return nullptr;
        std::shared_ptr<SchedJob> job = job_for(key, nt, n_cu, env);
        {
            std::unique_lock<std::mutex> jl(job->m);
            if (!job->done) {
                if (!wait) return nullptr;
                lock.unlock();
                job->cv.wait(jl, [&] { return job->done; });
                jl.unlock();
                lock.lock();
                if ((it = g_sched.find(key)) != g_sched.end()) { it->second.last_use = ++g_use_clock; return it->second.ok ? &it->second : nullptr; }
                jl.lock();
            }
            DagSchedule s;
            s.quarter_from = job->quarter_from; s.express_wgs = job->express_wgs; s.n_cu = job->n_cu;
            for (int g = 0; g < N_LISTS; ++g) { s.off[g] = job->off[g]; s.len[g] = job->len[g]; }
            if (job->valid && hipMalloc(&s.d_tasks, job->flat.size() * sizeof(unsigned long long)) == hipSuccess) {
                if (hipMemcpy(s.d_tasks, job->flat.data(), job->flat.size() * sizeof(unsigned long long), hipMemcpyHostToDevice) == hipSuccess) { s.n_tasks = (int)job->flat.size(); s.ok = true; }
                if (s.ok && !job->image.empty()) {
                    s.ok = hipMalloc(&s.d_image, job->image.size() * sizeof(int)) == hipSuccess && hipMalloc(&s.d_kmax, job->kmax.size() * sizeof(int)) == hipSuccess &&
                           hipMemcpy(s.d_image, job->image.data(), job->image.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
                           hipMemcpy(s.d_kmax, job->kmax.data(), job->kmax.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
                }
                if (!s.ok) {
                    if (s.d_tasks) (void)hipFree(s.d_tasks);
                    if (s.d_image) (void)hipFree(s.d_image);
                    if (s.d_kmax) (void)hipFree(s.d_kmax);
                    s.d_tasks = nullptr; s.d_image = nullptr; s.d_kmax = nullptr;
                }
            }
            if (!s.ok) (void)hipGetLastError();
            g_last_build_ms.store(job->build_ms);
            // a bounded cache: the least recently used lists go (nothing is in flight with them: a launch holds g_turn_mutex, not this one,
            // but its kernel reads d_tasks -- so only entries that were not used by the latest launches are dropped, and hipFree waits for the device)
            if (g_sched.size() >= SCHED_CACHE_MAX) {
                auto victim = g_sched.begin();
                for (auto c = g_sched.begin(); c != g_sched.end(); ++c) if (c->second.last_use < victim->second.last_use) victim = c;
                if (victim->second.d_tasks) (void)hipFree(victim->second.d_tasks);
                if (victim->second.d_image) (void)hipFree(victim->second.d_image);
                if (victim->second.d_kmax) (void)hipFree(victim->second.d_kmax);
                g_sched.erase(victim);
            }
            it = g_sched.emplace(key, s).first;
        }
        g_jobs.erase(key);
    }
    it->second.last_use = ++g_use_clock;
    return it->second.ok ? &it->second : nullptr;
}

constexpr size_t DAG_LDS_BYTES = (2 * (size_t)(LSH_DOUBLES + ISH_DOUBLES) + 8 + 2 * TILE) * sizeof(double);       // two teams' parked L_kk + mailboxes + y_k (the chain workgroup needs less)
static_assert(DAG_LDS_BYTES >= ((size_t)PACKED_TILE_DOUBLES + 2 * NB * NB + 16) * sizeof(double) && DAG_LDS_BYTES <= 160 * 1024, "LDS of the launch");

}  // namespace

size_t chol_dag_sync_ints(int nt) { return (size_t)dag_state_ints(nt); }

void chol_dag_init_device(int n_cu)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    { std::lock_guard<std::mutex> lock(g_sched_mutex); g_dev_cu[dev] = n_cu; }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_dag), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DAG_LDS_BYTES);
}

bool chol_dag_factor(double* S, double* y, double* x, int n_pad, const CholWorkspace& ws, double* ok, double* stall, hipStream_t st, const int** kmax_dev)
{
    const int nt = n_pad / TILE;
    if (kmax_dev) *kmax_dev = nullptr;
    if (nt < dag_min_tiles() || nt > 255) return false;
    static const bool sync_build = std::getenv("MAGE_CHOL_DAG_SYNC_BUILD") != nullptr;      // (tests that must see THIS schedule from the first factorisation on)
    const DagSchedule* s = get_schedule(nt, sync_build, ws.env_host);
    if (!s || s->n_cu < s->express_wgs + 1 + 2 * N_GROUPS) return false;
    int* state = ws.sync + 8;
    // Two of these launches from two streams of one process must not overlap: each wants every compute unit (one workgroup per unit),
    // the hardware deals the workgroups of both over the XCDs as units come free, and launch A holding all of XCD 3 while launch B holds
    // all of XCD 5 leaves A without servers for its group-5 list and B without servers for its group-3 list -- a circular wait that only
    // the bounded polls end (seen once in 25 000 steps of tests/test_soak_gpu.py).  So the launches of a device take turns: each waits for
    // the event recorded behind the previous one when that came from another stream.  A process that only ever uses ONE stream pays
    // nothing (an event record is a barrier packet: 6 us of idle stream between this launch and the backward solve): the events start
    // with the second stream seen, whose first launch also puts one behind whatever the first stream has queued so far.  (Launches
    // of OTHER processes cannot be ordered this way; there the bounded polls and the column-by-column fallback stay the answer.)
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> turn_lock(g_turn_mutex);
    Turnstile& turn = g_turn[dev];
    if (!turn.several && turn.last && turn.last != st) {
        if (!turn.ev && hipEventCreateWithFlags(&turn.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); turn.ev = nullptr; return false; }
        turn.several = true;
        turn.recorded = hipEventRecord(turn.ev, turn.last) == hipSuccess;       // (`last` is alive: chol_forget_stream clears it before a stream is destroyed)
        if (!turn.recorded) (void)hipGetLastError();
    }
    if (turn.several && turn.recorded && turn.last != st && hipStreamWaitEvent(st, turn.ev, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    // the state words: zero -- or, with an envelope, the image in which every panel of a structurally zero tile counts as applied
    if ((s->d_image ? hipMemcpyAsync(state, s->d_image, (size_t)dag_state_ints(nt) * sizeof(int), hipMemcpyDeviceToDevice, st)
                    : hipMemsetAsync(state, 0, (size_t)dag_state_ints(nt) * sizeof(int), st)) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (g_inject_stalls.load(std::memory_order_relaxed) > 0 && g_inject_stalls.fetch_sub(1) > 0) (void)hipMemsetAsync(state + D_INJECT, 1, sizeof(int), st);
    DagArgs a;
    a.S = S; a.y = y; a.x = x; a.Linv = ws.Linv; a.Lpub = ws.Linv + (size_t)nt * NBLK * NB * NB; a.part = a.Lpub + (size_t)nt * LPUB_TILE_DOUBLES; a.ok = ok; a.stall = stall;
    a.st = state; a.trace = ws.dbg; a.tasks = s->d_tasks; a.ld = n_pad; a.nt = nt; a.n_tasks = s->n_tasks; a.quarter_from = s->quarter_from; a.express_wgs = s->express_wgs;
    for (int g = 0; g < N_LISTS; ++g) { a.list_off[g] = s->off[g]; a.list_len[g] = s->len[g]; }
    hipLaunchKernelGGL(k_chol_dag, dim3(s->n_cu), dim3(DAG_THREADS), DAG_LDS_BYTES, st, a);
    if (hipGetLastError() != hipSuccess) return false;
    if (turn.several) {
        turn.recorded = hipEventRecord(turn.ev, st) == hipSuccess;
        if (!turn.recorded) (void)hipGetLastError();
    }
    turn.last = st;
    if (kmax_dev) *kmax_dev = s->d_kmax;
    return true;
}

// the order of a system is known (structure build): start building its lists now, if this size is the task graph's at all
void chol_dag_prefetch(int n_pad, const int* env_host)
{
    const int nt = n_pad / TILE;
    if (nt < dag_min_tiles() || nt > 255) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(g_sched_mutex);
    const int n_cu = device_cus(dev);
    const SchedKey key{ dev, nt, env_hash(env_host, nt) };
    if (n_cu > 0 && g_sched.find(key) == g_sched.end()) (void)job_for(key, nt, n_cu, env_host);
}

// blocks until the lists of this order are on the device (true) or cannot be had (false); how long the last build took on its thread
bool chol_dag_wait_schedule(int n_pad, double* build_ms, const int* env_host)
{
    const int nt = n_pad / TILE;
    bool ok = false;
    if (nt >= dag_min_tiles() && nt <= 255) ok = get_schedule(nt, true, env_host) != nullptr;
    if (build_ms) *build_ms = g_last_build_ms.load();
    return ok;
}

// a stream is about to be destroyed: the turnstile must not put an event behind it later
void chol_forget_stream(hipStream_t st)
{
    std::lock_guard<std::mutex> turn_lock(g_turn_mutex);
    for (auto& kv : g_turn) if (kv.second.last == st) kv.second.last = nullptr;
}

}  // namespace mage

// Test hook: the next n task-graph launches of this process abort as they do when a bounded wait runs out (*stall = 4): the caller's trial is
// re-run by the column-by-column launches and the process stays with them (tests/test_chol_gpu.py).
MAGE_EXPORT void mage_debug_chol_inject_stall(int n) { mage::g_inject_stalls.store(n, std::memory_order_relaxed); }

// Host-only view of the schedule for tests (tests/test_chol_schedule.py): the nine lists for nt tile columns on n_cu compute units, one
// behind the other (group_len[9] = their lengths; the last is the express list), and whether check_schedule accepts them.  Returns the total length (<= cap entries
// are written), negative when the lists fail the check.
MAGE_EXPORT int mage_debug_chol_schedule(int nt, int n_cu, int fuse_max, unsigned long long* out, int cap, int* quarter_from, int* group_len)
{
    if (nt < 2 || nt > 255 || n_cu < 4 + 1 + 2 * mage::N_GROUPS) return 0;
    const mage::Schedule sch = mage::build_schedule(nt, n_cu, fuse_max > 0 ? fuse_max : 8);
    if (quarter_from) *quarter_from = sch.quarter_from;
    int n = 0;
    for (int g = 0; g < mage::N_LISTS; ++g) {
        if (group_len) group_len[g] = (int)sch.lists[g].size();
        for (unsigned long long w : sch.lists[g]) { if (n < cap) out[n] = w; ++n; }
    }
    return mage::check_schedule(sch, nt) ? n : -n;
}

// The same for a system with an envelope (env[i] = first tile column of tile row i that can hold a non-zero): the lists skip every tile
// left of it; negative when the checker -- started from the envelope's initial progress -- rejects them.
MAGE_EXPORT int mage_debug_chol_schedule_env(int nt, int n_cu, int fuse_max, const int* env, unsigned long long* out, int cap, int* quarter_from, int* group_len)
{
    if (nt < 2 || nt > 255 || !env || n_cu < 4 + 1 + 2 * mage::N_GROUPS) return 0;
    for (int i = 0; i < nt; ++i) if (env[i] < 0 || env[i] > i) return 0;
    const mage::Schedule sch = mage::build_schedule(nt, n_cu, fuse_max > 0 ? fuse_max : 8, env);
    if (quarter_from) *quarter_from = sch.quarter_from;
    int n = 0;
    for (int g = 0; g < mage::N_LISTS; ++g) {
        if (group_len) group_len[g] = (int)sch.lists[g].size();
        for (unsigned long long w : sch.lists[g]) { if (n < cap) out[n] = w; ++n; }
    }
    return mage::check_schedule(sch, nt, env) ? n : -n;
}

