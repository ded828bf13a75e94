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

/* ------------------------------------------------------------------------------------------------
 * RadiusMatch  (Tracking/FeatureMatcher.cpp:294-446) + KeypointSpatialIndex::Query (Image/KeypointSpatialIndex.cpp:89-97):
 * per query, candidates = target keypoints inside the closed box [x-r, x+r] x [y-r, y+r] with
 * |octave_t - octave_q| * 100 <= 1; best Hamming distance below maxHammingDist + 1, "second best" = the previous best at
 * the time the best was last improved (NOT the true second smallest -- FeatureMatcher.cpp:423-434), accepted when
 * second - best > minHammingDifference; then a target claimed by several queries keeps only a strictly best one.
 * The candidate order comes from a boost R*-tree traversal in the reference (implementation-defined, boost is not
 * vendored): the canonical order here (and in the HIP path) is ASCENDING TARGET INDEX.  PARITY UNPINNED for that order.
 * ------------------------------------------------------------------------------------------------ */
typedef struct { float x, y, size, angle, response; int octave, class_id; } mto_keypoint;

MTO_API int mto_radius_match(const mto_keypoint* qk, int nq, const float* qpos_override /* nq x 2 or NULL */, const uint8_t* qmask,
                             const uint8_t* qdesc, const mto_keypoint* tk, int nt, const uint8_t* tmask, const uint8_t* tdesc,
                             float radius, int max_dist, int min_diff, mto_dmatch* out, int cap)
{
    mto_dmatch* almost = (mto_dmatch*)malloc(sizeof(mto_dmatch) * (size_t)(nq > 0 ? nq : 1));
    int na = 0;
    for (int q = 0; q < nq; ++q) {
        if (qmask && !qmask[q]) continue;
        const float px = qpos_override ? qpos_override[2 * q] : qk[q].x, py = qpos_override ? qpos_override[2 * q + 1] : qk[q].y;
        const float x0 = px - radius, x1 = px + radius, y0 = py - radius, y1 = py + radius;
        const float z0 = qk[q].octave * 100.0f - 1.0f, z1 = qk[q].octave * 100.0f + 1.0f;
        int best = max_dist + 1, second = 2147483647, train = -1;
        for (int t = 0; t < nt; ++t) {
            const float tz = tk[t].octave * 100.0f;
            if (!(tk[t].x >= x0 && tk[t].x <= x1 && tk[t].y >= y0 && tk[t].y <= y1 && tz >= z0 && tz <= z1)) continue;
            if (tmask && !tmask[t]) continue;
            const int d = mto_hamming256(qdesc + (size_t)q * 32, tdesc + (size_t)t * 32);
            if (d < best) { train = t; second = best; best = d; }
        }
        if (train != -1 && (second - best) > min_diff) {
            almost[na].queryIdx = q; almost[na].trainIdx = train; almost[na].imgIdx = 0; almost[na].distance = (float)best;
            ++na;
        }
    }
    int n = 0;
    if (na > 1) {
        float* b1 = (float*)malloc(sizeof(float) * (size_t)nt * 2);
        float* b2 = b1 + nt;
        for (int t = 0; t < nt; ++t) { b1[t] = 3.402823466e+38f; b2[t] = 3.402823466e+38f; }
        for (int i = 0; i < na; ++i) {
            const int t = almost[i].trainIdx; const float d = almost[i].distance;
            if (d < b1[t]) { b2[t] = b1[t]; b1[t] = d; }
            else if (d < b2[t]) b2[t] = d;
        }
        for (int i = 0; i < na; ++i) {
            const int t = almost[i].trainIdx;
            if (almost[i].distance == b1[t] && b1[t] < b2[t]) { if (n < cap) out[n] = almost[i]; ++n; }
        }
        free(b1);
    } else {
        for (int i = 0; i < na; ++i) { if (n < cap) out[n] = almost[i]; ++n; }
    }
    free(almost);
    return n;
}


/* ------------------------------------------------------------------------------------------------
 * IndexedMatch (Tracking/FeatureMatcher.cpp:192-292) with TrackMatch (:28-54).  The candidate lists are what
 * BaseBow::QueryFeatures / BaseFeatureMatcher::QueryFeatures returned for each descriptor (vocabulary index: out of scope),
 * handed over in CSR form and visited in the order given.  maxHamming = maxHammingDist + 1 with strict comparisons.
 * ------------------------------------------------------------------------------------------------ */
typedef struct { int idx, dist; } mto_track;

static void mto_track_match(const uint8_t* left, const uint8_t* right_descs, int idx_right, const uint8_t* right_mask, mto_track* best,
                            mto_track* second, int max_hamming)
{
    if (right_mask && !right_mask[idx_right]) return;                       /* :37 */
    const int d = mto_hamming256(left, right_descs + (size_t)idx_right * 32);
    if (d < max_hamming) {
        if (d < best->dist) { *second = *best; best->idx = idx_right; best->dist = d; }
        else if (d < second->dist) { second->idx = idx_right; second->dist = d; }
    }
}

MTO_API int mto_indexed_match(const uint8_t* descA, int nA, const uint8_t* maskA, const int32_t* cb_off, const int32_t* cb,
                              const uint8_t* descB, int nB, const uint8_t* maskB, const int32_t* ca_off, const int32_t* ca,
                              int max_dist, int min_diff, mto_dmatch* out, int cap)
{
    int cntA = 0, cntB = 0;
    for (int i = 0; i < nA; ++i) cntA += (!maskA || maskA[i]);
    for (int i = 0; i < nB; ++i) cntB += (!maskB || maskB[i]);
    if (cntA == 0 || cntB == 0) return 0;                                   /* :208 */
    const int max_hamming = max_dist + 1;                                   /* :210 */
    int n = 0;
    for (int a = 0; a < nA; ++a) {
        if (maskA && !maskA[a]) continue;
        mto_track best = { -1, max_hamming }, second = { -1, max_hamming };
        for (int k = cb_off[a]; k < cb_off[a + 1]; ++k) mto_track_match(descA + (size_t)a * 32, descB, cb[k], maskB, &best, &second, max_hamming);
        if (!(best.dist < max_hamming && (second.dist >= max_hamming || second.dist - best.dist >= min_diff))) continue;     /* :239-240 */
        const int b = best.idx;
        mto_track rb = { -1, max_hamming }, rs = { -1, max_hamming };
        for (int k = ca_off[b]; k < ca_off[b + 1]; ++k) mto_track_match(descB + (size_t)b * 32, descA, ca[k], maskA, &rb, &rs, max_hamming);
        if (rb.dist < max_hamming && rb.idx == a && (rs.dist >= max_hamming || rs.dist - rb.dist >= min_diff)) {              /* :269-271 */
            if (n < cap) { out[n].queryIdx = rb.idx; out[n].trainIdx = b; out[n].imgIdx = 0; out[n].distance = (float)rb.dist; }
            ++n;
        }
    }
    return n;
}


/* ------------------------------------------------------------------------------------------------
 * The vocabulary tree's leaf lookup, OnlineBow::FindLeafNode (Core/MAGESLAM/Source/BoW/OnlineBow.cpp:289-311): from the root, at every
 * level the child whose medoid descriptor is nearest in Hamming distance (GetDescriptorDistance, FeatureMatcher.cpp:453-504), strict
 * '<' in the order of the node's child list -- the FIRST of equally near children wins -- until a node without children.
 * The tree as flat arrays: node n's medoid is node_desc[32 n ..], its children are children[child_off[n] .. child_off[n + 1]); node 0
 * is the root (its descriptor is never compared).  Training (k-medoids, OnlineBow.cpp:325-500) and the node -> keyframe maps are the
 * caller's: out of scope.
 * ------------------------------------------------------------------------------------------------ */
MTO_API void mto_bow_find_leaf(const uint8_t* node_desc, const int32_t* child_off, const int32_t* children, const uint8_t* queries, int nq, int32_t* leaf)
{
    for (int q = 0; q < nq; ++q) {
        int cur = 0;
        while (child_off[cur] < child_off[cur + 1]) {                       /* :294 */
            int best_d = 0x7fffffff, next = cur;
            for (int k = child_off[cur]; k < child_off[cur + 1]; ++k) {     /* :299 */
                const int d = mto_hamming256(queries + (size_t)q * 32, node_desc + (size_t)children[k] * 32);
                if (d < best_d) { best_d = d; next = children[k]; }        /* :302-306 */
            }
            cur = next;
        }
        leaf[q] = cur;
    }
}

/* IndexedMatch with the candidate lists taken from the vocabulary as the reference does (FeatureMatcher.cpp:223-227, 253-257 ->
 * OnlineBow::QueryFeatures :115-132): the candidates of a descriptor are the features of the OTHER image filed under the leaf the
 * descriptor descends to -- feat_b[feat_b_off[leaf] ..) for a descriptor of A, feat_a[...] for one of B -- in the order they were filed. */
MTO_API int mto_indexed_match_bow(const uint8_t* node_desc, const int32_t* child_off, const int32_t* children,
                                  const uint8_t* descA, int nA, const uint8_t* maskA, const int32_t* feat_a_off, const int32_t* feat_a,
                                  const uint8_t* descB, int nB, const uint8_t* maskB, const int32_t* feat_b_off, const int32_t* feat_b,
                                  int max_dist, int min_diff, mto_dmatch* out, int cap)
{
    int cntA = 0, cntB = 0;
    for (int i = 0; i < nA; ++i) cntA += (!maskA || maskA[i]);
    for (int i = 0; i < nB; ++i) cntB += (!maskB || maskB[i]);
    if (cntA == 0 || cntB == 0) return 0;
    const int max_hamming = max_dist + 1;
    int n = 0;
    for (int a = 0; a < nA; ++a) {
        if (maskA && !maskA[a]) continue;
        int32_t la;
        mto_bow_find_leaf(node_desc, child_off, children, descA + (size_t)a * 32, 1, &la);
        mto_track best = { -1, max_hamming }, second = { -1, max_hamming };
        for (int k = feat_b_off[la]; k < feat_b_off[la + 1]; ++k) mto_track_match(descA + (size_t)a * 32, descB, feat_b[k], maskB, &best, &second, max_hamming);
        if (!(best.dist < max_hamming && (second.dist >= max_hamming || second.dist - best.dist >= min_diff))) continue;
        const int b = best.idx;
        int32_t lb;
        mto_bow_find_leaf(node_desc, child_off, children, descB + (size_t)b * 32, 1, &lb);
        mto_track rb = { -1, max_hamming }, rs = { -1, max_hamming };
        for (int k = feat_a_off[lb]; k < feat_a_off[lb + 1]; ++k) mto_track_match(descB + (size_t)b * 32, descA, feat_a[k], maskA, &rb, &rs, max_hamming);
        if (rb.dist < max_hamming && rb.idx == a && (rs.dist >= max_hamming || rs.dist - rb.dist >= min_diff)) {
            if (n < cap) { out[n].queryIdx = rb.idx; out[n].trainIdx = b; out[n].imgIdx = 0; out[n].distance = (float)rb.dist; }
            ++n;
        }
    }
    return n;
}
