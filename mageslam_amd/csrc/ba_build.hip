// ba_build.hip -- device build of the bundle-adjustment graph structure (see ba_build.h for what is built and why).
//
// The host twin is initialize_optimization's list code in ba_host.hip; the lists must come out identical, so every ORDER below
// is pinned by a key, never by which thread arrived first:
//   * hessian index of a camera / landmark index of a point: ascending camera / point index (exclusive scans)      [A.5]
//   * observations inside a landmark: free cameras ascending (ties: observation index), then fixed cameras by observation
//     index -- rank of the key (camera key << 32 | observation) inside the landmark's bucket
//   * W slots: a new slot where the free camera changes inside a landmark (scan over the ordered positions)
//   * a camera's observations (camE): ascending observation index; its slots (camS): ascending slot index -- stable splits by
//     camera: per-workgroup histograms, their column scan, and an in-wavefront rank by ballot
//   * contributions (slot_a, slot_b) of a block (i, j) of S: ascending slot_a -- stable split of row i by j, same machinery with
//     one bitmap per column camera in LDS (a lane can hold several columns)
// Integer atomics only add (counts): their results do not depend on order.
#include "ba_build.h"

#include <algorithm>

namespace mage {

namespace {

constexpr int WAVE = 64;
typedef unsigned long long u64;

__device__ __forceinline__ u64 lanemask_lt()
{
    const int lane = threadIdx.x & 63;
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

__device__ __forceinline__ u64 wave_scan_incl(u64 v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u64 t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// exclusive scan of one value per thread over a workgroup of T threads (T multiple of 64, <= 1024); *total = the sum
template <int T>
__device__ __forceinline__ u64 block_scan_excl(u64 v, u64* total, u64* sm /* T / 64 + 1 */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 incl = wave_scan_incl(v);
    __syncthreads();                       // sm may still be read by the previous call
    if (lane == 63) sm[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 run = 0;
        for (int w = 0; w < T / 64; ++w) { const u64 t = sm[w]; sm[w] = run; run += t; }
        sm[T / 64] = run;
    }
    __syncthreads();
    *total = sm[T / 64];
    return sm[wave] + incl - v;
}

// ---------------------------------------------------------------------------------------------------------------------
// Generic exclusive scan of 64-bit values produced by op.load(i) over i in [0, n); op.store(i, exclusive, value) consumes the
// result, op.finish(total) runs once.  One workgroup handles BUILD_SCAN_BLOCK consecutive elements, 8 per thread.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int SCAN_T = 256, SCAN_PER = BUILD_SCAN_BLOCK / SCAN_T;
constexpr int SCAN1_T = 1024;                          // the single-workgroup form: 8192 elements per round

template <typename Op>
__global__ __launch_bounds__(SCAN_T) void k_scan_reduce(Op op, int n, u64* partial)
{
    __shared__ u64 sm[SCAN_T / 64 + 1];
    const int base = blockIdx.x * BUILD_SCAN_BLOCK + threadIdx.x * SCAN_PER;
    u64 s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_PER; ++k) if (base + k < n) s += op.load(base + k);
    u64 total;
    (void)block_scan_excl<SCAN_T>(s, &total, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_scan_sums(u64* partial, int nb)
{
    __shared__ u64 sm[1024 / 64 + 1];
    u64 carry = 0;
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        const int i = b0 + threadIdx.x;
        const u64 v = i < nb ? partial[i] : 0;
        u64 total;
        const u64 ex = block_scan_excl<1024>(v, &total, sm);
        if (i < nb) partial[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) partial[nb] = carry;
}

// SINGLE: one workgroup of 1024 threads walks the whole range with a running carry (small inputs: one launch instead of three)
template <typename Op, bool SINGLE>
__global__ __launch_bounds__(SINGLE ? SCAN1_T : SCAN_T) void k_scan_apply(Op op, int n, const u64* partial, int nb)
{
    constexpr int T = SINGLE ? SCAN1_T : SCAN_T;
    __shared__ u64 sm[T / 64 + 1];
    u64 carry = SINGLE ? 0 : partial[blockIdx.x];
    const int b_first = SINGLE ? 0 : blockIdx.x, b_last = SINGLE ? nb : blockIdx.x + 1;
    for (int b = b_first; b < b_last; ++b) {
        const int base = b * (T * SCAN_PER) + threadIdx.x * SCAN_PER;
        u64 v[SCAN_PER];
        u64 s = 0;
#pragma unroll
        for (int k = 0; k < SCAN_PER; ++k) { v[k] = base + k < n ? op.load(base + k) : 0; s += v[k]; }
        u64 total;
        u64 run = carry + block_scan_excl<T>(s, &total, sm);
#pragma unroll
        for (int k = 0; k < SCAN_PER; ++k) {
            if (base + k < n) op.store(base + k, run, v[k]);
            run += v[k];
        }
        carry += total;
    }
    if (SINGLE) { if (threadIdx.x == 0) op.finish(carry); }
    else if (blockIdx.x == 0 && threadIdx.x == 0) op.finish(partial[nb]);
}

template <typename Op>
void launch_scan(const Op& op, int n, u64* tmp, hipStream_t st)
{
    const int nb1 = std::max(1, (n + SCAN1_T * SCAN_PER - 1) / (SCAN1_T * SCAN_PER));
    if (nb1 <= 3) {
        hipLaunchKernelGGL((k_scan_apply<Op, true>), dim3(1), dim3(SCAN1_T), 0, st, op, n, (const u64*)tmp, nb1);
        return;
    }
    const int nb = std::max(1, (n + BUILD_SCAN_BLOCK - 1) / BUILD_SCAN_BLOCK);
    hipLaunchKernelGGL((k_scan_reduce<Op>), dim3(nb), dim3(SCAN_T), 0, st, op, n, tmp);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, tmp, nb);
    hipLaunchKernelGGL((k_scan_apply<Op, false>), dim3(nb), dim3(SCAN_T), 0, st, op, n, (const u64*)tmp, nb);
}

// ---------------------------------------------------------------------------------------------------------------------
// Phase 1
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool obs_active(const BuildArgs& a, const ObsRecord& o)
{
    // set, not removed, not (camera fixed and points fixed): OptimizableGraph::Edge::allVerticesFixed
    return o.set && !o.removed && !(a.points_fixed && a.cam_fixed[o.cam]);
}

// Observations usually arrive by point (BuildDataForG2O walks the map points), so neighbouring lanes hold the same point: a run
// of equal keys among active lanes is counted (or given its bucket positions) by ONE atomic of its first lane.
struct LaneRun { int first, len; bool head; };
__device__ __forceinline__ LaneRun lane_run(uint32_t key, bool act)
{
    const int lane = threadIdx.x & 63;
    const uint32_t prev = __shfl_up(key, 1, 64);
    const bool prev_act = __shfl_up(act ? 1 : 0, 1, 64) != 0;
    const bool head = lane == 0 || !act || !prev_act || key != prev;
    const u64 H = __ballot(head);
    const u64 le = lanemask_lt() | (1ull << lane);
    const int first = 63 - __clzll(H & le);
    const u64 above = first == 63 ? 0ull : (H & ~((2ull << first) - 1ull));
    const int end = above ? __ffsll((long long)above) - 1 : 64;
    return LaneRun{ first, end - first, head };
}

constexpr int COUNT_T = 256, COUNT_PER = 4, COUNT_CAM_LDS = 8192;

__global__ __launch_bounds__(COUNT_T) void k_build_count(BuildArgs a)
{
    __shared__ int cam_cnt[COUNT_CAM_LDS];
    const bool lds_cams = a.n_cams <= COUNT_CAM_LDS;
    if (lds_cams) {
        for (int c = threadIdx.x; c < a.n_cams; c += COUNT_T) cam_cnt[c] = 0;
        __syncthreads();
    }
    int n_act = 0;
    for (int k = 0; k < COUNT_PER; ++k) {
        const int e = (blockIdx.x * COUNT_PER + k) * COUNT_T + threadIdx.x;
        bool act = false;
        ObsRecord o;
        o.cam = 0; o.pt = 0;
        if (e < a.n_obs) { o = a.obs[e]; act = obs_active(a, o); }
        if (act) {
            if (lds_cams) atomicAdd(&cam_cnt[o.cam], 1); else atomicAdd(&a.cam_deg[o.cam], 1);
        }
        const LaneRun r = lane_run(o.pt, act);
        if (act && r.head) atomicAdd(&a.pt_deg[o.pt], r.len);
        n_act += __popcll(__ballot(act));
    }
    if ((threadIdx.x & 63) == 0 && n_act) atomicAdd(&a.counts->n_L, n_act);
    if (lds_cams) {
        __syncthreads();
        for (int c = threadIdx.x; c < a.n_cams; c += COUNT_T) { const int t = cam_cnt[c]; if (t) atomicAdd(&a.cam_deg[c], t); }
    }
}

// hessian index of the cameras: free, and (observed or tethered or -- sharded maps -- any), ascending camera index
__global__ __launch_bounds__(1024) void k_build_cams(BuildArgs a)
{
    __shared__ u64 sm[1024 / 64 + 1];
    u64 carry = 0;
    for (int c0 = 0; c0 < a.n_cams; c0 += 1024) {
        const int c = c0 + threadIdx.x;
        bool in = false;
        if (c < a.n_cams) {
            const int deg = a.cam_deg[c] + (a.cam_extra_deg ? a.cam_extra_deg[c] : 0);
            in = !a.cam_fixed[c] && (deg > 0 || a.keep_all_free_cameras);
        }
        u64 total;
        const u64 ex = block_scan_excl<1024>(in ? 1 : 0, &total, sm);
        if (c < a.n_cams) {
            const int hc = (int)(carry + ex);
            a.cam2hc[c] = in ? hc : -1;
            if (in) a.hc2cam[hc] = c;
        }
        carry += total;
    }
    if (threadIdx.x == 0) a.counts->n_fc = (int)carry;
}

// landmarks = points with an active observation, ascending point index; lm_ptr = offsets of their observation runs
struct ScanPoints {
    BuildArgs a;
    __device__ u64 load(int i) const { const int d = a.pt_deg[i]; return d > 0 ? ((1ull << 32) | (u64)(uint32_t)d) : 0ull; }
    __device__ void store(int i, u64 ex, u64 v) const
    {
        if (v) {
            const int l = (int)(ex >> 32);
            a.pt2lm[i] = l; a.lm_pt[l] = i; a.lm_ptr[l] = (int)(ex & 0xffffffffu);
            a.pt_deg[i] = 0;                     // becomes the landmark's fill cursor
        } else a.pt2lm[i] = -1;
    }
    __device__ void finish(u64 total) const
    {
        const int nlm = (int)(total >> 32);
        a.counts->n_lm = nlm;
        a.lm_ptr[nlm] = (int)(total & 0xffffffffu);
    }
};

// every active observation into its landmark's run, in arrival order, with the key that orders the run
__global__ __launch_bounds__(256) void k_build_bucket(BuildArgs a)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    bool act = false;
    ObsRecord o;
    o.cam = 0; o.pt = 0;
    if (e < a.n_obs) {
        o = a.obs[e];
        act = obs_active(a, o);
        if (!act) a.where[e] = -1;
    }
    const LaneRun r = lane_run(o.pt, act);
    int base = 0;
    if (act && r.head) base = atomicAdd(&a.pt_deg[o.pt], r.len);
    base = __shfl(base, r.first, 64);
    if (!act) return;
    const int l = a.pt2lm[o.pt];
    const int pos = a.lm_ptr[l] + base + ((int)(threadIdx.x & 63) - r.first);
    const int hc = a.cam2hc[o.cam];
    a.bucket[pos] = ((u64)(uint32_t)(hc < 0 ? 0x7fffffff : hc) << 32) | (u64)(uint32_t)e;
}

// rank of every key inside its landmark's run = its position in landmark order; the observation record moves there
__global__ __launch_bounds__(256) void k_build_order(BuildArgs a)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= a.counts->n_L) return;
    const u64 key = a.bucket[p];
    const uint32_t e = (uint32_t)(key & 0xffffffffu);
    const ObsRecord o = a.obs[e];
    const int l = a.pt2lm[o.pt];
    const int b = a.lm_ptr[l], end = a.lm_ptr[l + 1];
    int rank = 0;
    for (int q = b; q < end; ++q) rank += a.bucket[q] < key ? 1 : 0;
    const int dst = b + rank;
    const uint32_t hck = (uint32_t)(key >> 32);
    a.L_edge[dst] = e; a.L_uv[dst] = make_float2(o.u, o.v); a.L_info[dst] = o.info; a.L_cam[dst] = o.cam; a.L_pt[dst] = o.pt;
    a.L_hc[dst] = hck == 0x7fffffffu ? -1 : (int)hck;
    a.L_lm[dst] = l;
    a.where[e] = dst;
}

// W slots: one per distinct (free camera, landmark) pair, numbered in landmark order; an observation of a fixed camera (or any
// observation when the points are fixed) has none
struct ScanSlots {
    BuildArgs a;
    __device__ u64 load(int p) const
    {
        if (p >= a.counts->n_L || a.points_fixed) return 0;
        const int hc = a.L_hc[p];
        if (hc < 0) return 0;
        const bool first = p == a.lm_ptr[a.L_lm[p]] || a.L_hc[p - 1] != hc;
        return (first ? (1ull << 32) : 0ull) | 1ull;
    }
    __device__ void store(int p, u64 ex, u64 v) const
    {
        if (p >= a.counts->n_L) return;
        const int l = a.L_lm[p];
        const int before = (int)(ex >> 32);                    // slots opened before this position
        if (p == a.lm_ptr[l]) a.lm_wptr[l] = before;
        if (v >> 32) { a.w_hc[before] = a.L_hc[p]; a.w_lm[before] = l; }
        a.L_slot[p] = v ? before + (int)(v >> 32) - 1 : -1;
    }
    __device__ void finish(u64 total) const
    {
        a.counts->n_w = (int)(total >> 32);
        a.counts->slot_obs = (int)(total & 0xffffffffu);
        a.lm_wptr[a.counts->n_lm] = (int)(total >> 32);
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// Stable split by camera (camE: observations in ascending observation index -> their positions; camS: slots ascending).
// Both splits share their launches: workgroups [0, nbE) take the observations, [nbE, nbE + nbS) the slots.  One wavefront per
// workgroup owns a contiguous chunk of the input.  A: per-chunk histogram.  B: per camera, the offset of every chunk's share --
// chunks are scanned in groups (B1: inside a group, many workgroups; B2: over the groups and over the cameras, one workgroup per
// split).  C: the chunk again, 64 items at a time in order; the rank of an item among the same camera's items of its batch comes
// from ballots over the bits of the camera index.  Sizes (n_fc, n_w) are read from *counts: the host has not seen them yet.
// ---------------------------------------------------------------------------------------------------------------------
struct SplitPlan { int nbE, nbS, groups, key_bits; };

struct SplitSide {
    int src, b, nb, n;          // which split, chunk index inside it, its chunks, its items
    int* hist; int* gtot;       // this split's histogram rows / group totals
};
__device__ __forceinline__ SplitSide split_side(const BuildArgs& a, const SplitPlan& p, int block, int n_fc)
{
    SplitSide s;
    s.src = block < p.nbE ? 0 : 1;
    s.b = s.src ? block - p.nbE : block;
    s.nb = s.src ? p.nbS : p.nbE;
    s.n = s.src ? a.counts->n_w : a.n_obs;
    s.hist = a.hist + (size_t)(s.src ? p.nbE : 0) * n_fc;
    s.gtot = a.hist + (size_t)(p.nbE + p.nbS) * n_fc + (size_t)(s.src ? p.groups : 0) * n_fc;
    return s;
}
__device__ __forceinline__ int split_key(const BuildArgs& a, int src, int i, int* val)
{
    if (src) { *val = i; return a.w_hc[i]; }
    const int p = a.where[i];
    *val = p;
    return p < 0 ? -1 : a.L_hc[p];
}
__device__ __forceinline__ void chunk_range(int n, int nb, int b, int* lo, int* hi)
{
    const int per = ((n + nb - 1) / nb + WAVE - 1) / WAVE * WAVE;
    *lo = min(n, b * per);
    *hi = min(n, *lo + per);
}

__global__ __launch_bounds__(WAVE) void k_split_count(BuildArgs a, SplitPlan p)
{
    extern __shared__ __attribute__((aligned(16))) int lds_i[];
    int* cnt = lds_i;
    const int n_fc = a.counts->n_fc;
    if (n_fc <= 0) return;
    const SplitSide s = split_side(a, p, blockIdx.x, n_fc);
    for (int c = threadIdx.x; c < n_fc; c += WAVE) cnt[c] = 0;
    __syncthreads();
    int lo, hi;
    chunk_range(s.n, s.nb, s.b, &lo, &hi);
    for (int i = lo + (int)threadIdx.x; i < hi; i += WAVE) {
        int val;
        const int k = split_key(a, s.src, i, &val);
        if (k >= 0) atomicAdd(&cnt[k], 1);
        if (s.src) a.w_end[i] = a.lm_wptr[a.w_lm[i] + 1];      // one past the last slot of the slot's landmark (the row kernels' chain is one load shorter)
    }
    __syncthreads();
    int* row = s.hist + (size_t)s.b * n_fc;
    for (int c = threadIdx.x; c < n_fc; c += WAVE) row[c] = cnt[c];
}

// B1: grid (camera tiles, groups, 2 splits): hist[b][c] -> exclusive prefix inside the group; gtot[g][c] = the group's total
__global__ __launch_bounds__(256) void k_split_group_scan(BuildArgs a, SplitPlan p)
{
    const int n_fc = a.counts->n_fc;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_fc) return;
    const int src = blockIdx.z, g = blockIdx.y;
    const int nb = src ? p.nbS : p.nbE;
    int* hist = a.hist + (size_t)(src ? p.nbE : 0) * n_fc;
    int* gtot = a.hist + (size_t)(p.nbE + p.nbS) * n_fc + (size_t)(src ? p.groups : 0) * n_fc;
    const int per = (nb + p.groups - 1) / p.groups;
    const int b0 = min(nb, g * per), b1 = min(nb, b0 + per);
    int run = 0;
    for (int b = b0; b < b1; ++b) { const int t = hist[(size_t)b * n_fc + c]; hist[(size_t)b * n_fc + c] = run; run += t; }
    gtot[(size_t)g * n_fc + c] = run;
}

// B2: one workgroup per split: gtot[g][c] -> offset of the group's first item of camera c; ptr[c] = camera offsets (n_fc + 1)
__global__ __launch_bounds__(1024) void k_split_base(BuildArgs a, SplitPlan p)
{
    __shared__ u64 sm[1024 / 64 + 1];
    const int n_fc = a.counts->n_fc;
    const int src = blockIdx.x;
    int* gtot = a.hist + (size_t)(p.nbE + p.nbS) * n_fc + (size_t)(src ? p.groups : 0) * n_fc;
    int* ptr = src ? a.camS_ptr : a.camE_ptr;
    u64 carry = 0;
    for (int c0 = 0; c0 < n_fc; c0 += 1024) {
        const int c = c0 + threadIdx.x;
        int tot = 0;
        if (c < n_fc) for (int g = 0; g < p.groups; ++g) tot += gtot[(size_t)g * n_fc + c];
        u64 total;
        const u64 ex = block_scan_excl<1024>((u64)tot, &total, sm);
        if (c < n_fc) {
            int run = (int)(carry + ex);
            ptr[c] = run;
            for (int g = 0; g < p.groups; ++g) { const int t = gtot[(size_t)g * n_fc + c]; gtot[(size_t)g * n_fc + c] = run; run += t; }
        }
        carry += total;
    }
    if (threadIdx.x == 0) ptr[max(n_fc, 0)] = (int)carry;
}

__global__ __launch_bounds__(WAVE) void k_split_scatter(BuildArgs a, SplitPlan p)
{
    extern __shared__ __attribute__((aligned(16))) int lds_i[];
    int* cnt = lds_i;
    const int n_fc = a.counts->n_fc;
    if (n_fc <= 0) return;
    const SplitSide s = split_side(a, p, blockIdx.x, n_fc);
    int* out = s.src ? a.camS : a.camE;
    const int per_g = (s.nb + p.groups - 1) / p.groups;
    const int* row = s.hist + (size_t)s.b * n_fc;
    const int* grow = s.gtot + (size_t)(s.b / per_g) * n_fc;
    for (int c = threadIdx.x; c < n_fc; c += WAVE) cnt[c] = row[c] + grow[c];
    __syncthreads();
    int lo, hi;
    chunk_range(s.n, s.nb, s.b, &lo, &hi);
    const u64 lt = lanemask_lt();
    for (int i0 = lo; i0 < hi; i0 += WAVE) {
        const int i = i0 + (int)threadIdx.x;
        int val = 0;
        const int k = i < hi ? split_key(a, s.src, i, &val) : -1;
        u64 m = __ballot(k >= 0);                        // lanes holding the same camera as this one
        for (int bit = 0; bit < p.key_bits; ++bit) {
            const u64 bm = __ballot((k >> bit) & 1);
            m &= ((k >> bit) & 1) ? bm : ~bm;
        }
        int base = 0;
        if (k >= 0) base = cnt[k];
        __syncthreads();                                 // every lane has read its counter
        if (k >= 0) {
            out[base + __popcll(m & lt)] = val;
            if ((m & lt) == 0) cnt[k] = base + __popcll(m);   // the group's first lane moves the counter on
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Rows of the reduced camera matrix.  Row i is built from the slots of camera i (camS: ascending slot = ascending landmark):
// every later slot b >= a of the same landmark is a camera j >= i.  k_build_row_count: the row's non-empty blocks and
// contributions.  k_build_row_fill: block offsets, and the contributions scattered into their blocks in ascending slot_a.
// A workgroup of NW wavefronts per row; wavefront w owns the w-th contiguous part of the row's slots.
// LDS: cnt[n_fc] | part[NW][n_fc] | bitmap[NW][n_fc] (u64).
// ---------------------------------------------------------------------------------------------------------------------
// The ROW_KC columns behind a slot are three 16-byte loads at a 4-byte aligned address (gfx950 takes unaligned global loads):
// one address per lane and instruction instead of twelve -- with 13 rows on 13 compute units the row kernels were bound by
// the address rate of those units, not by bandwidth.
typedef int int4_a4 __attribute__((ext_vector_type(4), aligned(4)));
static_assert(BUILD_ROW_KC == 12, "load_columns reads three int4");
__device__ __forceinline__ void load_columns(const int* w_hc, int sa, int* jj)
{
    const int4_a4* q = reinterpret_cast<const int4_a4*>(w_hc + sa);
    const int4_a4 x = q[0], y = q[1], z = q[2];
    jj[0] = x.x; jj[1] = x.y; jj[2] = x.z; jj[3] = x.w; jj[4] = y.x; jj[5] = y.y; jj[6] = y.z; jj[7] = y.w; jj[8] = z.x; jj[9] = z.y; jj[10] = z.z; jj[11] = z.w;
}
constexpr int ROW_KC = BUILD_ROW_KC;
constexpr int ROW_PF = 4;                 // slots (row_count) / 64-slot batches (row_fill) whose load chains are in flight together      // columns of a slot kept in registers across the phases of the scatter (longer tracks re-read)

__device__ __forceinline__ void row_part(int n_slots, int nw, int wave, int* b, int* e)
{
    const int per = ((n_slots + nw - 1) / nw + WAVE - 1) / WAVE * WAVE;
    *b = min(n_slots, wave * per);
    *e = min(n_slots, *b + per);
}

__global__ __launch_bounds__(1024) void k_build_row_count(BuildArgs a)
{
    extern __shared__ __attribute__((aligned(16))) int lds_i[];
    __shared__ u64 sm[1024 / 64 + 1];
    int* cnt = lds_i;
    const int n_fc = a.counts->n_fc;
    const int i = blockIdx.x, T = 1024;
    if (i >= n_fc) return;
    for (int c = threadIdx.x; c < n_fc; c += T) cnt[c] = 0;
    __syncthreads();
    const int s0 = a.camS_ptr[i], ns = a.camS_ptr[i + 1] - s0;
    // The chain slot -> end of its landmark -> its columns is two dependent loads; a thread's slots are fetched ROW_PF at a time so
    // that their chains are in flight together (the loop was a chain per slot: three round trips to L2 per thread on a local BA).
    for (int k0 = threadIdx.x; k0 < ns; k0 += ROW_PF * T) {
        int sa[ROW_PF], end[ROW_PF], jj[ROW_PF][ROW_KC];
#pragma unroll
        for (int u = 0; u < ROW_PF; ++u) sa[u] = k0 + u * T < ns ? a.camS[s0 + k0 + u * T] : 0;
#pragma unroll
        for (int u = 0; u < ROW_PF; ++u) {
            end[u] = k0 + u * T < ns ? a.w_end[sa[u]] : 0;     // (an idle slot reads slot 0's columns and uses none of them)
            load_columns(a.w_hc, sa[u], jj[u]);                // speculative: w_hc is padded by ROW_KC entries
        }
#pragma unroll
        for (int u = 0; u < ROW_PF; ++u) {
#pragma unroll
            for (int t = 0; t < ROW_KC; ++t) if (sa[u] + t < end[u]) atomicAdd(&cnt[jj[u][t]], 1);
            for (int sb = sa[u] + ROW_KC; sb < end[u]; ++sb) atomicAdd(&cnt[a.w_hc[sb]], 1);
        }
    }
    __syncthreads();
    u64 v = 0;
    for (int c = threadIdx.x; c < n_fc; c += T) {
        const int t = cnt[c];
        if (t > 0 || c == i) v += (1ull << 40) | (u64)t;      // the diagonal block of every free camera exists even when empty
    }
    u64 total;
    (void)block_scan_excl<1024>(v, &total, sm);
    if (threadIdx.x == 0) a.row[i] = total;
}

struct ScanRows {
    BuildArgs a;
    __device__ u64 load(int i) const { return i < a.counts->n_fc ? a.row[i] : 0ull; }
    __device__ void store(int i, u64 ex, u64) const { if (i < a.counts->n_fc) a.row[i] = ex; }
    __device__ void finish(u64 total) const
    {
        a.row[a.counts->n_fc] = total;
        a.counts->n_blk = (int)(total >> 40);
        a.counts->n_con = total & ((1ull << 40) - 1);
    }
};


template <int NW>
__global__ __launch_bounds__(NW * WAVE) void k_build_row_fill(BuildArgs a, int n_fc)
{
    extern __shared__ __attribute__((aligned(16))) int lds_i[];
    __shared__ u64 sm[NW + 1];
    int* cnt = lds_i;                                          // n_fc: contributions per column
    int* part = lds_i + n_fc;                                  // NW x n_fc: a wavefront's share, then its write cursor
    u64* bitmap = reinterpret_cast<u64*>(lds_i + (size_t)(1 + NW) * n_fc + ((1 + NW) * n_fc & 1));   // NW x n_fc, 8-byte aligned
    const int i = blockIdx.x, T = NW * WAVE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = threadIdx.x; c < (1 + NW) * n_fc; c += T) lds_i[c] = 0;
    for (int c = threadIdx.x; c < NW * n_fc; c += T) bitmap[c] = 0;
    __syncthreads();
    const int s0 = a.camS_ptr[i], ns = a.camS_ptr[i + 1] - s0;
    int pb, pe;
    row_part(ns, NW, wave, &pb, &pe);
    int* mine = part + (size_t)wave * n_fc;
    // Up to PF batches of 64 slots per wavefront (rows of up to PF * 64 * NW slots: every case measured) are fetched ONCE, all
    // chains in flight together, and kept in registers for both passes; longer rows re-read batch by batch.
    constexpr int PF = NW >= 16 ? 3 : ROW_PF;                                        // (1024 threads leave a wavefront 128 registers)
    const bool resident = ((ns + NW - 1) / NW + WAVE - 1) / WAVE <= PF;              // the same for every wavefront of the workgroup
    int rsa[PF], rend[PF], rjj[PF][ROW_KC];
    if (resident) {
#pragma unroll
        for (int u = 0; u < PF; ++u) rsa[u] = pb + u * WAVE + lane < pe ? a.camS[s0 + pb + u * WAVE + lane] : 0;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            rend[u] = pb + u * WAVE + lane < pe ? a.w_end[rsa[u]] : 0;
            load_columns(a.w_hc, rsa[u], rjj[u]);
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
#pragma unroll
            for (int t = 0; t < ROW_KC; ++t) {
                if (rsa[u] + t < rend[u]) atomicAdd(&mine[rjj[u][t]], 1); else rjj[u][t] = -1;
            }
            for (int sb = rsa[u] + ROW_KC; sb < rend[u]; ++sb) atomicAdd(&mine[a.w_hc[sb]], 1);
        }
    } else {
        for (int k = pb + lane; k < pe; k += WAVE) {
            const int sa = a.camS[s0 + k];
            const int end = a.w_end[sa];
            int jj[ROW_KC];
            load_columns(a.w_hc, sa, jj);
#pragma unroll
            for (int t = 0; t < ROW_KC; ++t) if (sa + t < end) atomicAdd(&mine[jj[t]], 1);
            for (int sb = sa + ROW_KC; sb < end; ++sb) atomicAdd(&mine[a.w_hc[sb]], 1);
        }
    }
    __syncthreads();
    // columns: totals, the wavefronts' shares turned into offsets inside the block, then block index / offset by a scan over j
    const u64 row_ex = a.row[i];
    const int blk0 = (int)(row_ex >> 40);
    const u64 con0 = row_ex & ((1ull << 40) - 1);
    // every thread owns a contiguous range of columns so that the scan runs in column order
    const int per = (n_fc + T - 1) / T;
    const int c_lo = min(n_fc, (int)threadIdx.x * per), c_hi = min(n_fc, c_lo + per);
    u64 v = 0;
    for (int c = c_lo; c < c_hi; ++c) {
        int run = 0;
        for (int w = 0; w < NW; ++w) { const int t = part[(size_t)w * n_fc + c]; part[(size_t)w * n_fc + c] = run; run += t; }
        cnt[c] = run;
        if (run > 0 || c == i) v += (1ull << 40) | (u64)run;
    }
    u64 total;
    u64 ex = block_scan_excl<NW * WAVE>(v, &total, sm);
    for (int c = c_lo; c < c_hi; ++c) {
        const int t = cnt[c];
        if (t > 0 || c == i) {
            const int blk = blk0 + (int)(ex >> 40);
            const u64 off = con0 + (ex & ((1ull << 40) - 1));
            a.blk_ptr[blk] = (int)off;
            a.blk_ij[blk] = make_int2(i, c);
            for (int w = 0; w < NW; ++w) part[(size_t)w * n_fc + c] += (int)off;
            ex += (1ull << 40) | (u64)t;
        }
    }
    if (i == n_fc - 1 && threadIdx.x == T - 1) a.blk_ptr[a.counts->n_blk] = (int)(a.row[n_fc] & ((1ull << 40) - 1));
    __syncthreads();
    // scatter: 64 slots of the part at a time, in order; lane -> (slot_a, every later slot of its landmark).  The rank of a
    // contribution inside its block among the batch = lower lanes that hold the same column: one bitmap per column.
    u64* bm = bitmap + (size_t)wave * n_fc;
    const u64 lt = lanemask_lt(), me = 1ull << lane;
    auto scatter_batch = [&](int sa, int end, int (&jj)[ROW_KC]) {
#pragma unroll
        for (int t = 0; t < ROW_KC; ++t) if (jj[t] >= 0) atomicOr(&bm[jj[t]], me);
        for (int sb = sa + ROW_KC; sb < end; ++sb) atomicOr(&bm[a.w_hc[sb]], me);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        u64 mm[ROW_KC];
#pragma unroll
        for (int t = 0; t < ROW_KC; ++t)
            if (jj[t] >= 0) {
                mm[t] = bm[jj[t]];
                a.con[(size_t)mine[jj[t]] + __popcll(mm[t] & lt)] = make_int2(sa, sa + t);
            } else mm[t] = 0;
        for (int sb = sa + ROW_KC; sb < end; ++sb) {
            const int j = a.w_hc[sb];
            a.con[(size_t)mine[j] + __popcll(bm[j] & lt)] = make_int2(sa, sb);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the column's first lane moves the cursor on ...
#pragma unroll
        for (int t = 0; t < ROW_KC; ++t) if (jj[t] >= 0 && (mm[t] & lt) == 0) mine[jj[t]] += __popcll(mm[t]);
        for (int sb = sa + ROW_KC; sb < end; ++sb) {
            const int j = a.w_hc[sb];
            const u64 m = bm[j];
            if ((m & lt) == 0) mine[j] += __popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ... and everybody clears what it set
#pragma unroll
        for (int t = 0; t < ROW_KC; ++t) if (jj[t] >= 0) bm[jj[t]] = 0;
        for (int sb = sa + ROW_KC; sb < end; ++sb) bm[a.w_hc[sb]] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    if (resident) {
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (pb + u * WAVE < pe) scatter_batch(rsa[u], rend[u], rjj[u]);          // uniform per wavefront
    } else {
        for (int k0 = pb; k0 < pe; k0 += WAVE) {
            const int k = k0 + lane;
            int sa = 0, end = 0;
            int jj[ROW_KC];
#pragma unroll
            for (int t = 0; t < ROW_KC; ++t) jj[t] = -1;
            if (k < pe) {
                sa = a.camS[s0 + k];
                end = a.w_end[sa];
                load_columns(a.w_hc, sa, jj);
            }
            const int nk = end - sa;
#pragma unroll
            for (int t = 0; t < ROW_KC; ++t) if (t >= nk) jj[t] = -1;
            scatter_batch(sa, end, jj);
        }
    }
}

// XCD runs of k_schur_block: block b belongs to XCD x = min(7, mid(b) * 8 / n_con), which does not decrease
// with b, so every XCD gets a contiguous run [first[x], first[x + 1]).
__device__ __forceinline__ int xcd_of_block(const int* blk_ptr, int b, u64 ncon)
{
    const u64 mid = ((u64)(uint32_t)blk_ptr[b] + (u64)(uint32_t)blk_ptr[b + 1]) / 2;
    if (!ncon) return 0;
    const u64 x = mid * 8 / ncon;
    return x > 7 ? 7 : (int)x;
}

__global__ __launch_bounds__(64) void k_build_xcd_runs(BuildArgs a)
{
    const int nblk = a.counts->n_blk;
    const u64 ncon = a.counts->n_con;
    const int x = threadIdx.x;
    int first = nblk;
    if (x <= 8) {
        // first block with xcd >= x (binary search over the monotone map)
        int lo = 0, hi = nblk;
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            if (xcd_of_block(a.blk_ptr, mid, ncon) >= x) hi = mid; else lo = mid + 1;
        }
        first = x == 8 ? nblk : lo;
        a.counts->xcd_first[x] = first;
    }
    const int next = __shfl_down(first, 1, 64);
    int len = x < 8 ? next - first : 0;
    for (int d = 1; d < 8; d <<= 1) len = max(len, __shfl_xor(len, d, 64));
    if (x == 0) a.counts->xcd_longest = len;
}

// slot -> block table: workgroup w of k_schur_block lands on XCD w % 8 and takes the next block of that XCD's run
__global__ __launch_bounds__(256) void k_build_blk_order(BuildArgs a, int n_slots)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_slots) return;
    const int x = s % 8, q = s / 8;
    const int first = a.counts->xcd_first[x], len = a.counts->xcd_first[x + 1] - first;
    a.blk_order[s] = q < len ? first + q : -1;
}

constexpr size_t ROW_LDS_LIMIT = 150 * 1024;
constexpr int SPLIT_GROUPS = 32;

size_t row_fill_lds(int nw, int n_fc)
{
    return ((size_t)(1 + nw) * n_fc + 1) * sizeof(int) + (size_t)nw * n_fc * sizeof(u64);
}

int split_blocks(int n_items, int n_fc)
{
    // a wavefront per ~256 items (four rounds of dependent loads); the histogram matrix (blocks x n_fc ints) stays below 16 MB
    const int by_items = std::max(1, (n_items + 255) / 256);
    const int by_hist = std::max(1, (int)(((size_t)4 << 20) / (size_t)std::max(n_fc, 1)));
    return std::max(1, std::min(std::min(by_items, by_hist), 2048));
}

SplitPlan split_plan(const BuildArgs& a, int n_fc_max)
{
    SplitPlan p;
    p.nbE = split_blocks(a.n_obs, n_fc_max);
    p.nbS = split_blocks(a.n_obs, n_fc_max);        // slots <= observations: the host has not seen their number yet
    p.groups = std::min(SPLIT_GROUPS, std::max(p.nbE, p.nbS));
    p.key_bits = 0;
    while ((1 << p.key_bits) < std::max(n_fc_max, 1)) ++p.key_bits;
    return p;
}

}  // namespace

size_t build_hist_ints(int n_obs, int n_fc_max)
{
    BuildArgs a{};
    a.n_obs = n_obs;
    const SplitPlan p = split_plan(a, n_fc_max);
    return (size_t)(p.nbE + p.nbS + 2 * p.groups) * (size_t)std::max(n_fc_max, 1) + 1;
}

int build_row_waves(int n_fc)
{
    int nw = 16;
    while (nw > 1 && row_fill_lds(nw, n_fc) > 64 * 1024) nw >>= 1;
    return nw;
}

void build_init_device()
{
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || done[dev]) return;
    done[dev] = true;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_build_row_fill<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ROW_LDS_LIMIT);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_build_row_fill<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ROW_LDS_LIMIT);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_build_row_fill<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ROW_LDS_LIMIT);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_build_row_fill<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ROW_LDS_LIMIT);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_build_row_fill<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ROW_LDS_LIMIT);
}

void build_launch_phase1(const BuildArgs& a, int n_fc_max, hipStream_t st)
{
    // counts | cam_deg | pt_deg are one allocation (ba_host.hip): one fill
    (void)hipMemsetAsync(a.counts, 0, build_zeroed_bytes(a.n_cams, a.n_pts), st);
    if (a.n_obs) hipLaunchKernelGGL(k_build_count, dim3((a.n_obs + COUNT_T * COUNT_PER - 1) / (COUNT_T * COUNT_PER)), dim3(COUNT_T), 0, st, a);
    hipLaunchKernelGGL(k_build_cams, dim3(1), dim3(1024), 0, st, a);
    launch_scan(ScanPoints{ a }, a.n_pts, a.scan_tmp, st);
    const int nbo = std::max(1, (a.n_obs + 255) / 256);
    if (a.n_obs) {
        hipLaunchKernelGGL(k_build_bucket, dim3(nbo), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_build_order, dim3(nbo), dim3(256), 0, st, a);
    }
    launch_scan(ScanSlots{ a }, a.n_obs, a.scan_tmp, st);
    if (n_fc_max <= 0) return;
    // per-camera views, then the rows of S counted: everything the host needs to size the remaining lists
    const SplitPlan p = split_plan(a, n_fc_max);
    const size_t lds = (size_t)n_fc_max * sizeof(int);
    hipLaunchKernelGGL(k_split_count, dim3(p.nbE + p.nbS), dim3(WAVE), lds, st, a, p);
    hipLaunchKernelGGL(k_split_group_scan, dim3((n_fc_max + 255) / 256, p.groups, 2), dim3(256), 0, st, a, p);
    hipLaunchKernelGGL(k_split_base, dim3(2), dim3(1024), 0, st, a, p);
    hipLaunchKernelGGL(k_split_scatter, dim3(p.nbE + p.nbS), dim3(WAVE), lds, st, a, p);
    hipLaunchKernelGGL(k_build_row_count, dim3(n_fc_max), dim3(1024), lds, st, a);
    launch_scan(ScanRows{ a }, n_fc_max, a.scan_tmp, st);
}

void build_launch_row_fill(const BuildArgs& a, int n_fc, bool want_xcd_runs, hipStream_t st)
{
    if (n_fc <= 0) return;
    const int nw = build_row_waves(n_fc);
    const size_t lds = row_fill_lds(nw, n_fc);
    switch (nw) {
    case 16: hipLaunchKernelGGL(k_build_row_fill<16>, dim3(n_fc), dim3(16 * WAVE), lds, st, a, n_fc); break;
    case 8: hipLaunchKernelGGL(k_build_row_fill<8>, dim3(n_fc), dim3(8 * WAVE), lds, st, a, n_fc); break;
    case 4: hipLaunchKernelGGL(k_build_row_fill<4>, dim3(n_fc), dim3(4 * WAVE), lds, st, a, n_fc); break;
    case 2: hipLaunchKernelGGL(k_build_row_fill<2>, dim3(n_fc), dim3(2 * WAVE), lds, st, a, n_fc); break;
    default: hipLaunchKernelGGL(k_build_row_fill<1>, dim3(n_fc), dim3(WAVE), lds, st, a, n_fc); break;
    }
    if (want_xcd_runs) hipLaunchKernelGGL(k_build_xcd_runs, dim3(1), dim3(64), 0, st, a);
}

void build_launch_blk_order(const BuildArgs& a, int n_blk_slots, hipStream_t st)
{
    if (n_blk_slots <= 0) return;
    hipLaunchKernelGGL(k_build_blk_order, dim3((n_blk_slots + 255) / 256), dim3(256), 0, st, a, n_blk_slots);
}

}  // namespace mage
