// match_kernels.hip -- 256-bit Hamming brute-force two-way matcher (gfx950); integer work, no MFMA.
//
// Replaces Tracking/FeatureMatcher.cpp:61-190 (Match: two cv::BFMatcher::radiusMatch calls + ratio-by-difference +
// mutual check) and :448-504 (GetDescriptorDistance).  One workgroup per image pair: both descriptor sets are staged in
// LDS as 4 x u64 per descriptor (<= 2 x 14 KB at 440 features), every thread owns query rows and walks the
// other set with xor + v_bcnt; best / second-best / in-radius count are kept per row, then the mutual check and
// an ordered compaction emit cv::DMatch records in ascending query order.  Batched over pairs on blockIdx.x.
#include <cstdlib>
#include <cstring>

#include "orb_kernels.h"

namespace mage {
namespace {

constexpr int MT = 512;           // threads per pair
constexpr int LDS_DESC = 2048;    // descriptors per side that fit the LDS staging (2 x 2048 x 32 B = 128 KiB)

// (T must be a pointer whose address space the compiler can see at the call -- the LDS array itself or a global pointer, never a
// runtime choice between the two: a generic pointer makes every read a flat load with a full wait in front of the distance, one
// descriptor per ~180 cycles instead of per ~50.  Four descriptors are fetched ahead of the four distances.)
__device__ __forceinline__ void best_of_row(const ulonglong4 q, const ulonglong4* __restrict__ T, int nt, int max_dist, int min_diff, int& best, int& bd)
{
    int d1 = 1 << 30, d2 = 1 << 30, t1 = -1, cnt = 0;
    auto dist = [&](const ulonglong4 v) { return __popcll(q.x ^ v.x) + __popcll(q.y ^ v.y) + __popcll(q.z ^ v.z) + __popcll(q.w ^ v.w); };
    auto take = [&](int d, int t) {
        if (d <= max_dist) {
            ++cnt;
            if (d < d1) { d2 = d1; d1 = d; t1 = t; }
            else if (d < d2) d2 = d;
        }
    };
    int t = 0;
    for (; t + 4 <= nt; t += 4) {
        const ulonglong4 v0 = T[t], v1 = T[t + 1], v2 = T[t + 2], v3 = T[t + 3];
        const int e0 = dist(v0), e1 = dist(v1), e2 = dist(v2), e3 = dist(v3);
        if (min(min(e0, e1), min(e2, e3)) <= max_dist) { take(e0, t); take(e1, t + 1); take(e2, t + 2); take(e3, t + 3); }   // rare: one test for four rows
    }
    for (; t < nt; ++t) take(dist(T[t]), t);
    if (cnt == 0 || (cnt > 1 && (d2 - d1) < min_diff)) { best = -1; bd = 0; }
    else { best = t1; bd = d1; }
}

__global__ __launch_bounds__(MT) void k_match(const uint8_t* __restrict__ descA, const int* __restrict__ countsA, int capA,
                                              const uint8_t* __restrict__ descB, const int* __restrict__ countsB, int capB,
                                              int max_dist, int min_diff, int* __restrict__ scratch, mage_dmatch* __restrict__ out, int cap_out,
                                              int* __restrict__ counts, int use_lds)
{
    extern __shared__ ulonglong4 sm[];
    __shared__ int wave_cnt[MT / 64];
    __shared__ int base_s;
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nA = countsA[p], nB = countsB[p];
    const ulonglong4* A = reinterpret_cast<const ulonglong4*>(descA + (size_t)p * capA * 32);
    const ulonglong4* B = reinterpret_cast<const ulonglong4*>(descB + (size_t)p * capB * 32);
    int* f = scratch + (size_t)p * (capA + capB) * 2;    // f[q] best train, f[capA + q] distance
    int* g = f + 2 * capA;                              // g[t] best query
    mage_dmatch* o = out + (size_t)p * cap_out;
    if (nA == 0 || nB == 0) { if (tid == 0) counts[p] = 0; return; }
    if (use_lds) {
        for (int i = tid; i < nA; i += MT) sm[i] = A[i];
        for (int i = tid; i < nB; i += MT) sm[nA + i] = B[i];
        __syncthreads();
        for (int q = tid; q < nA; q += MT) { int b, d; best_of_row(sm[q], sm + nA, nB, max_dist, min_diff, b, d); f[q] = b; f[capA + q] = d; }
        for (int t = tid; t < nB; t += MT) { int b, d; best_of_row(sm[nA + t], sm, nA, max_dist, min_diff, b, d); g[t] = b; }
    } else {
        for (int q = tid; q < nA; q += MT) { int b, d; best_of_row(A[q], B, nB, max_dist, min_diff, b, d); f[q] = b; f[capA + q] = d; }
        for (int t = tid; t < nB; t += MT) { int b, d; best_of_row(B[t], A, nA, max_dist, min_diff, b, d); g[t] = b; }
    }
    __threadfence_block();
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int q0 = 0; q0 < nA; q0 += MT) {
        const int q = q0 + tid;
        int t = -1;
        if (q < nA) { t = f[q]; if (t >= 0 && g[t] != q) t = -1; }
        const unsigned long long bal = __ballot(t >= 0);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (t >= 0 && off + before < cap_out) {
            mage_dmatch m = { q, t, -1, (float)f[capA + q] };
            o[off + before] = m;
        }
        __syncthreads();
        if (tid == 0) { int s = 0; for (int w = 0; w < MT / 64; ++w) s += wave_cnt[w]; base_s += s; }
        __syncthreads();
    }
    if (tid == 0) counts[p] = base_s;
}

// ---------------------------------------------------------------------------------------------
// The same Match for FEW pairs (what the tracker submits: one or two per frame): one workgroup per pair leaves 255 compute units
// idle for the 2 x nA x nB distances, so a pair is spread over ceil(capA / 64) + ceil(capB / 64) workgroups.  A workgroup owns 64
// rows of one direction -- one per lane -- and stages the other set in LDS; its eight wavefronts walk interleaved eighths of that
// set (every lane of a wavefront reads the same descriptor: a broadcast), their (best, second best, in-radius count) are merged
// through LDS, and the workgroup that finishes LAST for its pair runs the mutual check and the ordered compaction.  Same results
// as k_match bit for bit: a tie for the best goes to the lower index, as the sequential scan does.
// ---------------------------------------------------------------------------------------------
constexpr int MR = 64, MW = 8;      // rows per workgroup, wavefronts (each walks every MW-th descriptor of the other set)

struct RowBest { int d1, t1, d2, cnt; };

__device__ __forceinline__ RowBest merge_best(const RowBest a, const RowBest b)
{
    RowBest r;
    const bool b_wins = b.d1 < a.d1 || (b.d1 == a.d1 && b.t1 < a.t1);
    r.d2 = min(min(a.d2, b.d2), max(a.d1, b.d1));
    r.d1 = min(a.d1, b.d1);
    r.t1 = b_wins ? b.t1 : a.t1;
    r.cnt = a.cnt + b.cnt;
    return r;
}

__global__ __launch_bounds__(MR * MW) void k_match_rows(const uint8_t* __restrict__ descA, const int* __restrict__ countsA, int capA,
                                                        const uint8_t* __restrict__ descB, const int* __restrict__ countsB, int capB,
                                                        int max_dist, int min_diff, int* __restrict__ scratch, mage_dmatch* __restrict__ out, int cap_out,
                                                        int* __restrict__ counts, int* __restrict__ done, int groupsA, int use_lds)
{
    extern __shared__ ulonglong4 sm[];
    __shared__ int4 part[MW][MR];
    __shared__ int wave_cnt[MW];
    __shared__ int base_s, is_last;
    const int p = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nA = countsA[p], nB = countsB[p];
    const ulonglong4* A = reinterpret_cast<const ulonglong4*>(descA + (size_t)p * capA * 32);
    const ulonglong4* B = reinterpret_cast<const ulonglong4*>(descB + (size_t)p * capB * 32);
    int* f = scratch + (size_t)p * (capA + capB) * 2;    // f[q] best train, f[capA + q] distance
    int* g = f + 2 * capA;                              // g[t] best query
    mage_dmatch* o = out + (size_t)p * cap_out;
    const bool dirA = (int)blockIdx.x < groupsA;
    const int row0 = (dirA ? (int)blockIdx.x : (int)blockIdx.x - groupsA) * MR;
    const int nQ = dirA ? nA : nB, nT = dirA ? nB : nA;
    if (nA > 0 && nB > 0 && row0 < nQ) {
        const ulonglong4* Q = dirA ? A : B;
        const ulonglong4* Tg = dirA ? B : A;
        if (use_lds) {
            for (int i = tid; i < nT; i += MR * MW) sm[i] = Tg[i];
            __syncthreads();
        }
        const int row = row0 + lane;
        RowBest st = { 1 << 30, 1 << 30, 1 << 30, 0 };
        if (row < nQ) {
            const ulonglong4 q = Q[row];
            // (the walk is written once per address space: through a pointer that is "LDS or global" every read is a flat load with a
            // full wait in front of the distance; four descriptors are fetched ahead of the four distances, one threshold test for the four)
            auto walk = [&](const ulonglong4* __restrict__ T) {
                auto dist = [&](const ulonglong4 v) { return __popcll(q.x ^ v.x) + __popcll(q.y ^ v.y) + __popcll(q.z ^ v.z) + __popcll(q.w ^ v.w); };
                auto take = [&](int d, int t) {
                    if (d <= max_dist) {
                        ++st.cnt;
                        if (d < st.d1) { st.d2 = st.d1; st.d1 = d; st.t1 = t; }
                        else if (d < st.d2) st.d2 = d;
                    }
                };
                int t = wave;
                for (; t + 3 * MW < nT; t += 4 * MW) {
                    const ulonglong4 v0 = T[t], v1 = T[t + MW], v2 = T[t + 2 * MW], v3 = T[t + 3 * MW];
                    const int e0 = dist(v0), e1 = dist(v1), e2 = dist(v2), e3 = dist(v3);
                    if (min(min(e0, e1), min(e2, e3)) <= max_dist) { take(e0, t); take(e1, t + MW); take(e2, t + 2 * MW); take(e3, t + 3 * MW); }
                }
                for (; t < nT; t += MW) take(dist(T[t]), t);
            };
            if (use_lds) walk(sm); else walk(Tg);
        }
        part[wave][lane] = make_int4(st.d1, st.t1, st.d2, st.cnt);
        __syncthreads();
        if (wave == 0 && row < nQ) {
#pragma unroll
            for (int w = 1; w < MW; ++w) { const int4 v = part[w][lane]; st = merge_best(st, RowBest{ v.x, v.y, v.z, v.w }); }
            const bool none = st.cnt == 0 || (st.cnt > 1 && (st.d2 - st.d1) < min_diff);
            if (dirA) { f[row] = none ? -1 : st.t1; f[capA + row] = none ? 0 : st.d1; }
            else g[row] = none ? -1 : st.t1;
        }
    }
    // The last workgroup of the pair to get here sees every row's result.  Only wavefront 0 stored results, so ONE release fence in
    // its thread 0 (write-back of this XCD's L2, the pairs' workgroups sit on different XCDs) publishes them before the count; a
    // fence in every wavefront would cost eight write-backs per workgroup.
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const int prev = __hip_atomic_fetch_add(&done[p], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = prev == (int)gridDim.x - 1;
        if (is_last) __hip_atomic_store(&done[p], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // ready for the next launch
        base_s = 0;
    }
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (nA == 0 || nB == 0) { if (tid == 0) counts[p] = 0; return; }
    for (int q0 = 0; q0 < nA; q0 += MR * MW) {
        const int q = q0 + tid;
        int t = -1;
        if (q < nA) { t = f[q]; if (t >= 0 && g[t] != q) t = -1; }
        const unsigned long long bal = __ballot(t >= 0);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (t >= 0 && off + before < cap_out) {
            mage_dmatch m = { q, t, -1, (float)f[capA + q] };
            o[off + before] = m;
        }
        __syncthreads();
        if (tid == 0) { int sum = 0; for (int w = 0; w < MW; ++w) sum += wave_cnt[w]; base_s += sum; }
        __syncthreads();
    }
    if (tid == 0) counts[p] = base_s;
}

// ---------------------------------------------------------------------------------------------
// RadiusMatch (Tracking/FeatureMatcher.cpp:294-446; candidates of KeypointSpatialIndex::Query in ascending target
// index -- the canonical order, see include/mage_match.h).  One workgroup per (query set, target set) problem:
// thread per query for the windowed best / "previous best" search, then the per-target uniqueness pass and an
// ordered compaction.  With ~440 targets a linear scan with the box test beats building any spatial index.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MT) void k_radius_match(const mage_keypoint* __restrict__ qk, int nq, const float2* __restrict__ qpos,
                                                     const uint8_t* __restrict__ qmask, const uint8_t* __restrict__ qdesc,
                                                     const mage_keypoint* __restrict__ tk, int nt, const uint8_t* __restrict__ tmask,
                                                     const uint8_t* __restrict__ tdesc, float radius, int max_dist, int min_diff,
                                                     int* __restrict__ scratch /* nq x 2 + nt x 2 ints */, mage_dmatch* __restrict__ out, int cap,
                                                     int* __restrict__ count)
{
#pragma clang fp contract(off)          // box bounds are compared with keypoint coordinates: plain IEEE operations, never an FMA
    __shared__ int wave_cnt[MT / 64];
    __shared__ int base_s, n_almost;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* a_train = scratch;            // nq : best target of query q or -1
    int* a_dist = scratch + nq;        // nq
    int* b1 = scratch + 2 * nq;        // nt : smallest claimed distance per target (Hamming distances are integers: atomicMin is exact)
    int* cnt = b1 + nt;                // nt : number of claims equal to that minimum
    const ulonglong4* Q = reinterpret_cast<const ulonglong4*>(qdesc);
    const ulonglong4* T = reinterpret_cast<const ulonglong4*>(tdesc);
    if (tid == 0) { base_s = 0; n_almost = 0; }
    for (int t = tid; t < nt; t += MT) { b1[t] = 0x7fffffff; cnt[t] = 0; }
    __syncthreads();
    int mine = 0;
    for (int q = tid; q < nq; q += MT) {
        int train = -1, best = max_dist + 1, second = 0x7fffffff;
        if (!qmask || qmask[q]) {
            const mage_keypoint k = qk[q];
            const float px = qpos ? qpos[q].x : k.x, py = qpos ? qpos[q].y : k.y;
            const float x0 = px - radius, x1 = px + radius, y0 = py - radius, y1 = py + radius;
            const float z0 = (float)k.octave * 100.0f - 1.0f, z1 = (float)k.octave * 100.0f + 1.0f;
            const ulonglong4 qd = Q[q];
            for (int t = 0; t < nt; ++t) {
                const mage_keypoint c = tk[t];
                const float tz = (float)c.octave * 100.0f;
                if (!(c.x >= x0 && c.x <= x1 && c.y >= y0 && c.y <= y1 && tz >= z0 && tz <= z1)) continue;
                if (tmask && !tmask[t]) continue;
                const ulonglong4 v = T[t];
                const int d = __popcll(qd.x ^ v.x) + __popcll(qd.y ^ v.y) + __popcll(qd.z ^ v.z) + __popcll(qd.w ^ v.w);
                if (d < best) { train = t; second = best; best = d; }
            }
            if (!(train != -1 && (second - best) > min_diff)) train = -1;
        }
        a_train[q] = train; a_dist[q] = best;
        if (train >= 0) ++mine;
    }
    atomicAdd(&n_almost, mine);
    __threadfence_block();
    __syncthreads();
    // With more than one accepted query (FeatureMatcher.cpp:344-373) a target keeps a claim only if it is the strictly
    // smallest one: claim == minimum and no second claim with that same distance (bestDistance < secondBestDistance).
    const bool filter = n_almost > 1;
    if (filter) {
        for (int q = tid; q < nq; q += MT) if (a_train[q] >= 0) atomicMin(&b1[a_train[q]], a_dist[q]);
        __threadfence_block();
        __syncthreads();
        for (int q = tid; q < nq; q += MT) { const int t = a_train[q]; if (t >= 0 && a_dist[q] == b1[t]) atomicAdd(&cnt[t], 1); }
        __threadfence_block();
    }
    __syncthreads();
    for (int q0 = 0; q0 < nq; q0 += MT) {
        const int q = q0 + tid;
        int t = -1;
        if (q < nq) {
            t = a_train[q];
            if (t >= 0 && filter && !(a_dist[q] == b1[t] && cnt[t] == 1)) t = -1;
        }
        const unsigned long long bal = __ballot(t >= 0);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (t >= 0 && off + before < cap) {
            mage_dmatch m = { q, t, 0, (float)a_dist[q] };
            out[off + before] = m;
        }
        __syncthreads();
        if (tid == 0) { int s2 = 0; for (int w = 0; w < MT / 64; ++w) s2 += wave_cnt[w]; base_s += s2; }
        __syncthreads();
    }
    if (tid == 0) *count = base_s;
}

// ---------------------------------------------------------------------------------------------
// IndexedMatch (Tracking/FeatureMatcher.cpp:192-292, TrackMatch :28-54).  Candidates come from a vocabulary index
// (QueryFeatures: out of scope) as CSR lists and are visited in the order given.  One workgroup per problem, thread per
// A descriptor: forward pass over its candidates, then -- if it passed -- the reverse pass over the candidates of the B
// descriptor it chose; accepted pairs are compacted in ascending A index.  Integer work, a few dozen 32-byte gathers per
// thread: latency-bound at 440 features, there is nothing to tile.
// ---------------------------------------------------------------------------------------------
struct Track { int idx, dist; };

__device__ __forceinline__ void track_match(const ulonglong4& left, const ulonglong4* __restrict__ right, int idx, const uint8_t* __restrict__ mask,
                                            Track& best, Track& second, int max_hamming)
{
    if (mask && !mask[idx]) return;
    const ulonglong4 v = right[idx];
    const int d = __popcll(left.x ^ v.x) + __popcll(left.y ^ v.y) + __popcll(left.z ^ v.z) + __popcll(left.w ^ v.w);
    if (d < max_hamming) {
        if (d < best.dist) { second = best; best.idx = idx; best.dist = d; }
        else if (d < second.dist) { second.idx = idx; second.dist = d; }
    }
}

// The vocabulary's leaf lookup (OnlineBow::FindLeafNode, BoW/OnlineBow.cpp:289-311) for a batch of descriptors: from the root, at every level
// to the child whose medoid is nearest in Hamming distance -- strict '<' in child-list order: the first of equally near children wins.
// Round 6: a walk is a chain of dependent trips to memory (five levels of child offsets -> child index -> medoid: ~40 loads one after the
// other on one thread, 18.8 us for 880 descriptors on four workgroups), so the tree goes up as a WALK TABLE -- per position k of the
// concatenated child lists one 48-byte entry {child node, the child's own range of positions, the child's medoid} (bow_walk_fill, built
// on the host beside the validation) -- and a level is ONE trip: a query is walked by 16 lanes, lane l rating positions k0 + l, k0 + l + 16, ...,
// key = distance << 16 | position in the list (the minimum is the nearest child, the first of equals), four row rotations reduce it over
// the 16 lanes, the winner's entry names the next range.  Four queries per wavefront, a wavefront per workgroup: 880 descriptors are
// 220 workgroups on as many compute units.  Integer work: bit-exact.  (Not staged through LDS: a workgroup walks four queries, and
// fetching the top two levels into LDS first is itself the trip it would save.)
__global__ __launch_bounds__(64) void k_bow_find_leaf(const BowWalkEntry* __restrict__ W, int root_k0, int root_k1, const uint8_t* __restrict__ queries, int nq, int* __restrict__ leaf)
{
    const int lane = threadIdx.x, sub = lane & 15, q = blockIdx.x * 4 + (lane >> 4);
    const bool live = q < nq;
    const ulonglong4 d = reinterpret_cast<const ulonglong4*>(queries)[live ? q : 0];
    int cur = 0, k0 = live ? root_k0 : 0, k1 = live ? root_k1 : 0;
    while (__builtin_amdgcn_ballot_w64(k0 < k1)) {
        unsigned key = 0xffffffffu;
        int c = cur, ck0 = 0, ck1 = 0;
        for (int k = k0 + sub; k < k1; k += 16) {
            const int4 hd = *reinterpret_cast<const int4*>(&W[k]);
            const ulonglong2 m0 = *reinterpret_cast<const ulonglong2*>(W[k].medoid), m1 = *reinterpret_cast<const ulonglong2*>(W[k].medoid + 2);
            const unsigned dist = (unsigned)(__popcll(d.x ^ m0.x) + __popcll(d.y ^ m0.y) + __popcll(d.z ^ m1.x) + __popcll(d.w ^ m1.y));
            const unsigned kk = (dist << 16) | (unsigned)(k - k0);
            if (kk < key) { key = kk; c = hd.x; ck0 = hd.y; ck1 = hd.z; }
        }
        // minimum over the 16 lanes of the query (row rotations by 8, 4, 2, 1: every lane ends with the row's minimum)
        unsigned best = key;
        best = min(best, (unsigned)__builtin_amdgcn_update_dpp((int)best, (int)best, 0x128, 0xf, 0xf, false));
        best = min(best, (unsigned)__builtin_amdgcn_update_dpp((int)best, (int)best, 0x124, 0xf, 0xf, false));
        best = min(best, (unsigned)__builtin_amdgcn_update_dpp((int)best, (int)best, 0x122, 0xf, 0xf, false));
        best = min(best, (unsigned)__builtin_amdgcn_update_dpp((int)best, (int)best, 0x121, 0xf, 0xf, false));
        // the lane that holds it (keys of one query are distinct: they carry the position) hands its entry to the other fifteen
        const unsigned long long win = __builtin_amdgcn_ballot_w64(key == best && key != 0xffffffffu);
        const int src = (lane & 48) + __builtin_ctz((unsigned)((win >> (lane & 48)) & 0xffffu) | 0x10000u);
        const int wc = __shfl(c, src & 63, 64), wk0 = __shfl(ck0, src & 63, 64), wk1 = __shfl(ck1, src & 63, 64);
        if (k0 < k1) { cur = wc; k0 = wk0; k1 = wk1; }
    }
    if (live && sub == 0) leaf[q] = cur;
}

// leafA / leafB (IndexedMatch through the vocabulary, mage_match_indexed_bow): when given, descriptor a's candidate list is the list of the
// LEAF it descended to -- cb_off / cb are then indexed by node (the other image's features filed under that node) instead of by descriptor.
__global__ __launch_bounds__(MT) void k_indexed_match(const uint8_t* __restrict__ descA, int nA, const uint8_t* __restrict__ maskA,
                                                      const int* __restrict__ cb_off, const int* __restrict__ cb,
                                                      const uint8_t* __restrict__ descB, const uint8_t* __restrict__ maskB,
                                                      const int* __restrict__ ca_off, const int* __restrict__ ca, int max_dist, int min_diff,
                                                      mage_dmatch* __restrict__ out, int cap, int* __restrict__ count,
                                                      const int* __restrict__ leafA, const int* __restrict__ leafB)
{
    __shared__ int wave_cnt[MT / 64];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const ulonglong4* A = reinterpret_cast<const ulonglong4*>(descA);
    const ulonglong4* B = reinterpret_cast<const ulonglong4*>(descB);
    const int max_hamming = max_dist + 1;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int a0 = 0; a0 < nA; a0 += MT) {
        const int a = a0 + tid;
        int train = -1, dist = 0;
        if (a < nA && (!maskA || maskA[a])) {
            const ulonglong4 da = A[a];
            Track best = { -1, max_hamming }, second = { -1, max_hamming };
            const int la = leafA ? leafA[a] : a;
            for (int k = cb_off[la]; k < cb_off[la + 1]; ++k) track_match(da, B, cb[k], maskB, best, second, max_hamming);
            if (best.dist < max_hamming && (second.dist >= max_hamming || second.dist - best.dist >= min_diff)) {
                const int b = best.idx;
                const ulonglong4 db = B[b];
                Track rb = { -1, max_hamming }, rs = { -1, max_hamming };
                const int lb = leafB ? leafB[b] : b;
                for (int k = ca_off[lb]; k < ca_off[lb + 1]; ++k) track_match(db, A, ca[k], maskA, rb, rs, max_hamming);
                if (rb.dist < max_hamming && rb.idx == a && (rs.dist >= max_hamming || rs.dist - rb.dist >= min_diff)) { train = b; dist = rb.dist; }
            }
        }
        const unsigned long long bal = __ballot(train >= 0);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (train >= 0 && off + before < cap) {
            mage_dmatch m = { a, train, 0, (float)dist };
            out[off + before] = m;
        }
        __syncthreads();
        if (tid == 0) { int s2 = 0; for (int w = 0; w < MT / 64; ++w) s2 += wave_cnt[w]; base_s += s2; }
        __syncthreads();
    }
    if (tid == 0) *count = base_s;
}

}  // namespace

void indexed_match_launch(const uint8_t* descA, int nA, const uint8_t* maskA, const int* cb_off, const int* cb, const uint8_t* descB,
                          const uint8_t* maskB, const int* ca_off, const int* ca, int max_dist, int min_diff, mage_dmatch* out, int cap, int* count,
                          hipStream_t st, const int* leafA, const int* leafB)
{
    hipLaunchKernelGGL(k_indexed_match, dim3(1), dim3(MT), 0, st, descA, nA, maskA, cb_off, cb, descB, maskB, ca_off, ca, max_dist, min_diff, out, cap, count, leafA, leafB);
}

void bow_find_leaf_launch(const BowWalkEntry* walk, int root_k0, int root_k1, const uint8_t* queries, int nq, int* leaf, hipStream_t st)
{
    if (nq > 0) hipLaunchKernelGGL(k_bow_find_leaf, dim3((nq + 3) / 4), dim3(64), 0, st, walk, root_k0, root_k1, queries, nq, leaf);
}

// the walk table of a (validated) tree: entry k = position k of the concatenated child lists
void bow_walk_fill(const uint8_t* node_descriptors, const int32_t* child_offsets, const int32_t* children, int n_nodes, BowWalkEntry* dst)
{
    const int nch = child_offsets[n_nodes];
    for (int k = 0; k < nch; ++k) {
        const int c = children[k];
        dst[k].child = c; dst[k].k0 = child_offsets[c]; dst[k].k1 = child_offsets[c + 1]; dst[k].pad = 0;
        std::memcpy(dst[k].medoid, node_descriptors + 32 * (size_t)c, 32);
    }
}

void radius_match_launch(const mage_keypoint* qk, int nq, const float2* qpos, const uint8_t* qmask, const uint8_t* qdesc, const mage_keypoint* tk,
                         int nt, const uint8_t* tmask, const uint8_t* tdesc, float radius, int max_dist, int min_diff, int* scratch,
                         mage_dmatch* out, int cap, int* count, hipStream_t st)
{
    hipLaunchKernelGGL(k_radius_match, dim3(1), dim3(MT), 0, st, qk, nq, qpos, qmask, qdesc, tk, nt, tmask, tdesc, radius, max_dist, min_diff,
                       scratch, out, cap, count);
}

void match_init_device()
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_match), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_DESC * 32);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_match_rows), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_DESC * 32);
}

// `done`: one int per pair, zero on entry (the kernel leaves it so).  Few pairs are spread over many workgroups each; from a few
// hundred pairs on the one-workgroup-per-pair kernel fills the device by itself and stages each set once.
void match_launch(int n_pairs, const uint8_t* descA, const int* countsA, int capA, const uint8_t* descB, const int* countsB, int capB,
                  int max_dist, int min_diff, int* scratch, mage_dmatch* out, int cap_out, int* counts, int* done, hipStream_t st)
{
    constexpr int split_below = 256;
    const int groupsA = (capA + MR - 1) / MR, groupsB = (capB + MR - 1) / MR;
    if (n_pairs < split_below && groupsA + groupsB > 0) {
        const int cap_t = capA > capB ? capA : capB;          // the staged set is the OTHER side's
        const int rows_lds = cap_t <= 2 * LDS_DESC ? 1 : 0;
        hipLaunchKernelGGL(k_match_rows, dim3(groupsA + groupsB, n_pairs), dim3(MR * MW), rows_lds ? (size_t)cap_t * 32 : 0, st, descA, countsA, capA, descB, countsB,
                           capB, max_dist, min_diff, scratch, out, cap_out, counts, done, groupsA, rows_lds);
        return;
    }
    const int use_lds = (capA + capB) <= 2 * LDS_DESC ? 1 : 0;
    const size_t lds = use_lds ? (size_t)(capA + capB) * 32 : 0;
    hipLaunchKernelGGL(k_match, dim3(n_pairs), dim3(MT), lds, st, descA, countsA, capA, descB, countsB, capB, max_dist, min_diff, scratch, out,
                       cap_out, counts, use_lds);
}

}  // namespace mage
