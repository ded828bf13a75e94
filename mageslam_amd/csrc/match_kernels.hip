// match_kernels.hip -- 256-bit Hamming brute-force two-way matcher (gfx950); integer work, no MFMA.
//
// Replaces Tracking/FeatureMatcher.cpp:61-190 (Match: two cv::BFMatcher::radiusMatch calls + ratio-by-difference +
// mutual check) and :448-504 (GetDescriptorDistance).  One workgroup per image pair: both descriptor sets are staged in
// LDS as 4 x u64 per descriptor (<= 2 x 14 KB at 440 features), every thread owns query rows and walks the
// other set with xor + v_bcnt; best / second-best / in-radius count are kept per row, then the mutual check and
// an ordered compaction emit cv::DMatch records in ascending query order.  Batched over pairs on blockIdx.x.
#include "orb_kernels.h"

namespace mage {
namespace {

constexpr int MT = 512;           // threads per pair
constexpr int LDS_DESC = 2048;    // descriptors per side that fit the LDS staging (2 x 2048 x 32 B = 128 KiB)

__device__ __forceinline__ void best_of_row(const ulonglong4 q, const ulonglong4* __restrict__ T, int nt, int max_dist, int min_diff, int& best, int& bd)
{
    int d1 = 1 << 30, d2 = 1 << 30, t1 = -1, cnt = 0;
    for (int t = 0; t < nt; ++t) {
        const ulonglong4 v = T[t];
        const int d = __popcll(q.x ^ v.x) + __popcll(q.y ^ v.y) + __popcll(q.z ^ v.z) + __popcll(q.w ^ v.w);
        if (d <= max_dist) {
            ++cnt;
            if (d < d1) { d2 = d1; d1 = d; t1 = t; }
            else if (d < d2) d2 = d;
        }
    }
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
    const ulonglong4* LA = A;
    const ulonglong4* LB = B;
    if (use_lds) {
        for (int i = tid; i < nA; i += MT) sm[i] = A[i];
        for (int i = tid; i < nB; i += MT) sm[nA + i] = B[i];
        __syncthreads();
        LA = sm; LB = sm + nA;
    }
    for (int q = tid; q < nA; q += MT) { int b, d; best_of_row(LA[q], LB, nB, max_dist, min_diff, b, d); f[q] = b; f[capA + q] = d; }
    for (int t = tid; t < nB; t += MT) { int b, d; best_of_row(LB[t], LA, nA, max_dist, min_diff, b, d); g[t] = b; }
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

}  // namespace

void match_init_device()
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_match), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS_DESC * 32);
}

void match_launch(int n_pairs, const uint8_t* descA, const int* countsA, int capA, const uint8_t* descB, const int* countsB, int capB,
                  int max_dist, int min_diff, int* scratch, mage_dmatch* out, int cap_out, int* counts, hipStream_t st)
{
    const int use_lds = (capA + capB) <= 2 * LDS_DESC ? 1 : 0;
    const size_t lds = use_lds ? (size_t)(capA + capB) * 32 : 0;
    hipLaunchKernelGGL(k_match, dim3(n_pairs), dim3(MT), lds, st, descA, countsA, capA, descB, countsB, capB, max_dist, min_diff, scratch, out,
                       cap_out, counts, use_lds);
}

}  // namespace mage
