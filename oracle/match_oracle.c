/*
 * match_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Restatement of the reference's brute-force two-way descriptor matcher:
 *   Core/MAGESLAM/Source/Tracking/FeatureMatcher.cpp:448-504  GetDescriptorDistance(Slow)
 *   Core/MAGESLAM/Source/Tracking/FeatureMatcher.cpp:61-190   Match
 * plus the behaviour of cv::BFMatcher(NORM_HAMMING).radiusMatch it calls (OpenCV 3.4.0, external;
 * SURVEY.md appendix A.7/A.8): keep train descriptors with distance <= maxDistance, sorted by distance.
 *
 * Canonical choices where the reference is unspecified or out of bounds (same in the HIP path):
 *  * bestBackwardsMatch is sized by the number of B descriptors and "no entry" is -1 (the reference sizes it by
 *    the COMPACTED row count and indexes it by original queryIdx: latent out-of-bounds);
 *  * with minHammingDifference == 0 a tie for the best distance is resolved towards the lowest train index
 *    (std::sort order is unspecified there); with minHammingDifference >= 1 ties reject the query, so the
 *    result is independent of sort order.
 *
 * PARITY UNPINNED by the reference (no tests).  Pinned by tests/test_match_oracle.py: SWAR popcount vs
 * bin(x ^ y).count("1"), and the matcher against an independent numpy restatement.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MTO_API __attribute__((visibility("default")))

typedef struct { int queryIdx, trainIdx, imgIdx; float distance; } mto_dmatch;    /* cv::DMatch */

/* FeatureMatcher.cpp:489-500: 8 x 32-bit SWAR popcount */
MTO_API int mto_hamming256(const uint8_t* a, const uint8_t* b)
{
    int result = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t x, y;
        memcpy(&x, a + 4 * i, 4); memcpy(&y, b + 4 * i, 4);
        uint32_t bits = x ^ y;
        bits = bits - ((bits >> 1) & 0x55555555u);
        bits = (bits & 0x33333333u) + ((bits >> 2) & 0x33333333u);
        result += (int)((((bits + (bits >> 4)) & 0x0F0F0F0Fu) * 0x01010101u) >> 24);
    }
    return result;
}

/* best / second-best within radius for every query row; best index -1 when the query is rejected */
static void one_way(const uint8_t* Q, int nq, const uint8_t* T, int nt, int max_dist, int min_diff, int* best, int* best_d)
{
    for (int q = 0; q < nq; ++q) {
        int d1 = 1 << 30, d2 = 1 << 30, t1 = -1, cnt = 0;
        for (int t = 0; t < nt; ++t) {
            int d = mto_hamming256(Q + (size_t)q * 32, T + (size_t)t * 32);
            if (d > max_dist) continue;
            ++cnt;
            if (d < d1) { d2 = d1; d1 = d; t1 = t; }
            else if (d < d2) d2 = d;
        }
        if (cnt == 0 || (cnt > 1 && (d2 - d1) < min_diff)) { best[q] = -1; best_d[q] = 0; }
        else { best[q] = t1; best_d[q] = d1; }
    }
}

/* Match on already gathered descriptor sets A (nA x 32) and B (nB x 32); indices refer to those sets. */
MTO_API int mto_match(const uint8_t* A, int nA, const uint8_t* B, int nB, int max_dist, int min_diff, mto_dmatch* out, int cap)
{
    if (nA == 0 || nB == 0) return 0;
    int* f = (int*)malloc(sizeof(int) * (size_t)nA * 2);
    int* g = (int*)malloc(sizeof(int) * (size_t)nB * 2);
    one_way(A, nA, B, nB, max_dist, min_diff, f, f + nA);
    one_way(B, nB, A, nA, max_dist, min_diff, g, g + nB);
    int n = 0;
    for (int q = 0; q < nA; ++q) {
        int t = f[q];
        if (t >= 0 && g[t] == q) {
            if (n < cap) { out[n].queryIdx = q; out[n].trainIdx = t; out[n].imgIdx = -1; out[n].distance = (float)f[nA + q]; }
            ++n;
        }
    }
    free(f); free(g);
    return n;
}
