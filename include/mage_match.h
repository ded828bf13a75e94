/*
 * mage_match.h -- C ABI of the MI355X brute-force Hamming matcher (libmageslam_hip.so).
 *
 * Drop-in boundary for the reference's free functions:
 *     Core/MAGESLAM/Source/Tracking/FeatureMatcher.h:100-109 / FeatureMatcher.cpp:61-190   Match
 *     Core/MAGESLAM/Source/Tracking/FeatureMatcher.cpp:448-504                            GetDescriptorDistance
 * Match = two-way radius match (cv::BFMatcher NORM_HAMMING, distance <= maxHammingDist), reject a query when
 * second-best minus best < minHammingDifference, keep mutual best pairs, emit cv::DMatch in ascending query order.
 * Masks are applied by gathering (FeatureMatcher.cpp:88-110) and indices are mapped back (:160-163); both are
 * done by mage_match_masked below so a caller can pass its std::vector<bool>-derived byte masks directly.
 * RadiusMatch (FeatureMatcher.cpp:294-446) is mage_match_radius below; IndexedMatch (:192-292) is
 * mage_match_indexed: its candidate lists are whatever the caller's vocabulary index returned; mage_match_indexed_bow looks them up in the
 * vocabulary tree on the device (mage_bow_find_leaf_batch = OnlineBow::FindLeafNode).  Training the tree stays with the caller.
 */
#ifndef MAGE_MATCH_H
#define MAGE_MATCH_H

#include <stddef.h>
#include <stdint.h>

#include "mage_ba.h"   /* mage_status */
#include "mage_orb.h"  /* mage_keypoint */

#ifdef __cplusplus
extern "C" {
#endif

/* cv::DMatch */
typedef struct mage_dmatch {
    int   queryIdx, trainIdx, imgIdx;   /* imgIdx = -1 */
    float distance;
} mage_dmatch;

/* GetDescriptorDistance on two 32-byte descriptors in host memory (host-side helper: one pair is not GPU work). */
int mage_hamming256(const uint8_t* d0, const uint8_t* d1);

typedef struct mage_matcher mage_matcher;
mage_status mage_matcher_create(int device, mage_matcher** out);
void        mage_matcher_destroy(mage_matcher* h);

/* Match on two gathered descriptor sets (host memory, nA x 32 and nB x 32 bytes).  Indices in the result refer
 * to the given sets.  Up to `capacity` matches are written; *count is the full number found. */
mage_status mage_match_bf(mage_matcher* h, const uint8_t* descA, int nA, const uint8_t* descB, int nB, int max_hamming_dist,
                          int min_hamming_difference, mage_dmatch* out, int capacity, int* count);

/* The reference's full signature: descriptor sets + byte masks (non-zero = use); result indices are the ORIGINAL
 * indices into descA / descB.  maskA / maskB may be NULL (= all). */
mage_status mage_match_masked(mage_matcher* h, const uint8_t* descA, int nDescA, const uint8_t* maskA, const uint8_t* descB,
                              int nDescB, const uint8_t* maskB, int max_hamming_dist, int min_hamming_difference,
                              mage_dmatch* out, int capacity, int* count);

/* Batched over independent pairs.  Pair p matches set A_p (countsA[p] descriptors at descA + p * capA * 32) against
 * B_p likewise; results go to out + p * cap_out, counts[p].  Host pointers. */
mage_status mage_match_bf_batch(mage_matcher* h, int n_pairs, const uint8_t* descA, const int* countsA, int capA,
                                const uint8_t* descB, const int* countsB, int capB, int max_hamming_dist,
                                int min_hamming_difference, mage_dmatch* out, int cap_out, int* counts);

/* Same with every buffer already in the handle's HBM (e.g. straight from mage_orb_detect_batch_device);
 * out / counts are device pointers owned by the handle, valid until its next call. */
mage_status mage_match_bf_batch_device(mage_matcher* h, int n_pairs, const uint8_t* descA_dev, const int* countsA_dev, int capA,
                                       const uint8_t* descB_dev, const int* countsB_dev, int capB, int max_hamming_dist,
                                       int min_hamming_difference, int cap_out, const mage_dmatch** out_dev, const int** counts_dev);

/* RadiusMatch, multi-query form (Tracking/FeatureMatcher.cpp:294-378, built on the single-query form :386-446 and on
 * KeypointSpatialIndex::Query, Image/KeypointSpatialIndex.cpp:89-97).  For every unmasked query keypoint: candidates are the
 * target keypoints inside the closed box [x - radius, x + radius] x [y - radius, y + radius] around the query position
 * (query_position_override, nQ x 2 floats, replaces the keypoint's own position when non-NULL) on the same octave; the
 * best Hamming distance must be <= max_hamming_dist and beat the PREVIOUS best at its last improvement by more than
 * min_hamming_difference (the reference's running "second best", not the true one); a target claimed by several queries
 * keeps only a strictly smallest claim.  Results: cv::DMatch(queryIdx, trainIdx, imgIdx = 0, distance), ascending query.
 * The reference visits candidates in boost R*-tree order (implementation-defined); this implementation and its oracle use
 * ASCENDING TARGET INDEX -- the accept/reject outcome can depend on that order, parity with the reference is unpinned there.
 * All pointers are host pointers; masks may be NULL (= all). */
mage_status mage_match_radius(mage_matcher* h, const mage_keypoint* query_keypoints, int nQ, const float* query_position_override,
                              const uint8_t* query_mask, const uint8_t* query_descriptors, const mage_keypoint* target_keypoints, int nT,
                              const uint8_t* target_mask, const uint8_t* target_descriptors, float radius, int max_hamming_dist,
                              int min_hamming_difference, mage_dmatch* out, int capacity, int* count);

/* IndexedMatch (Tracking/FeatureMatcher.cpp:192-292; TrackMatch :28-54): the two-way best / second-best test of Match, but every
 * descriptor is compared only with the candidates a vocabulary index returned for it (BaseBow::QueryFeatures(descriptor, keyframe) or
 * BaseFeatureMatcher::QueryFeatures(descriptor), in the BoW directory -- out of scope: the caller passes the lists).
 *   cand_b_offsets[nA + 1], cand_b[]  CSR: candidates (indices into B) of each A descriptor, in the order QueryFeatures returned them
 *   cand_a_offsets[nB + 1], cand_a[]  CSR: candidates (indices into A) of each B descriptor, used for the reverse check
 * Forward: an unmasked A descriptor keeps its best unmasked candidate when best < max_hamming_dist + 1 and either no second
 * candidate is below that limit or second - best >= min_hamming_difference (strict '<' updates in list order: the first of
 * equal distances wins, a duplicated candidate becomes its own second best).  Reverse: the chosen B descriptor must choose
 * that A descriptor back under the same test.  Results: cv::DMatch(queryIdx = index in A, trainIdx = index in B, 0, distance),
 * ascending queryIdx.  Returns without matches when either mask selects nothing (:208).  Masks may be NULL (= all); all
 * pointers are host pointers; candidate indices outside the other image are rejected with MAGE_ERR_INVALID_ARGUMENT. */
mage_status mage_match_indexed(mage_matcher* h, const uint8_t* descriptors_a, int nA, const uint8_t* mask_a, const int32_t* cand_b_offsets,
                               const int32_t* cand_b, const uint8_t* descriptors_b, int nB, const uint8_t* mask_b, const int32_t* cand_a_offsets,
                               const int32_t* cand_a, int max_hamming_dist, int min_hamming_difference, mage_dmatch* out, int capacity, int* count);

/* The vocabulary tree of BoW/OnlineBow (Core/MAGESLAM/Source/BoW/OnlineBow.h: m_nodes, each a medoid descriptor + childrenIDs) as flat arrays:
 * node n's medoid is node_descriptors[32 n ..], its children are children[child_offsets[n] .. child_offsets[n + 1]) IN THE ORDER of the
 * reference's childrenIDs; node 0 is the root (its descriptor is never compared); a node without children is a leaf.  Every child index
 * must be larger than its parent's (nodes are appended as they are created, OnlineBow.cpp:560-640): anything else is MAGE_ERR_INVALID_ARGUMENT.
 * Training (k-medoids, OnlineBow.cpp:325-500) and the node -> keyframe maps are the caller's. */
typedef struct mage_bow_tree {
    const uint8_t* node_descriptors;   /* n_nodes x 32 bytes */
    const int32_t* child_offsets;      /* n_nodes + 1 */
    const int32_t* children;           /* child_offsets[n_nodes] node indices */
    int32_t n_nodes;
} mage_bow_tree;

/* OnlineBow::FindLeafNode (BoW/OnlineBow.cpp:289-311) for n descriptors at once: from the root, at every level the child whose medoid is
 * nearest in Hamming distance (GetDescriptorDistance), strict '<' in child-list order -- the first of equally near children wins -- until
 * a node without children; leaf_ids[i] = that node.  Host pointers. */
mage_status mage_bow_find_leaf_batch(mage_matcher* h, const mage_bow_tree* tree, const uint8_t* descriptors, int n, int32_t* leaf_ids);
/* Keeps a validated copy of the tree on the device (OnlineBow's tree changes once, when its training completes, BoW/OnlineBow.cpp:77-92; a lookup per
 * frame then moves 28 KB of descriptors instead of the 355-KB tree): mage_bow_find_leaf_batch and mage_match_indexed_bow called with
 * tree == NULL use it.  tree == NULL here forgets it.  Call again whenever the tree has changed. */
mage_status mage_bow_set_tree(mage_matcher* h, const mage_bow_tree* tree);

/* IndexedMatch (Tracking/FeatureMatcher.cpp:192-292) with its candidate lists looked up the way the reference does (:223-227, :253-257 ->
 * OnlineBow::QueryFeatures, BoW/OnlineBow.cpp:115-132): the candidates of a descriptor are the features of the OTHER image filed under the
 * leaf the descriptor descends to, in filing order.
 *   leaf_features_b_offsets[n_nodes + 1], leaf_features_b[]   per node the indices (into B) of image B's features filed under it
 *   leaf_features_a_offsets[n_nodes + 1], leaf_features_a[]   the same for image A (the reverse check)
 * (= m_NodeKeyframeMap[node][keyframe].indexes of the two keyframes; only leaves hold features).  Everything else -- masks, limits, the
 * strict updates in list order, the output records -- as mage_match_indexed.  One upload, two launches (the leaves of all nA + nB
 * descriptors, then the two-way test), one read-back: no host trip per descriptor. */
mage_status mage_match_indexed_bow(mage_matcher* h, const mage_bow_tree* tree, const uint8_t* descriptors_a, int nA, const uint8_t* mask_a,
                                   const int32_t* leaf_features_a_offsets, const int32_t* leaf_features_a, const uint8_t* descriptors_b, int nB,
                                   const uint8_t* mask_b, const int32_t* leaf_features_b_offsets, const int32_t* leaf_features_b, int max_hamming_dist,
                                   int min_hamming_difference, mage_dmatch* out, int capacity, int* count);

/* HIP-event time of the most recent batched call's kernel, in milliseconds. */
mage_status mage_matcher_last_kernel_ms(const mage_matcher* h, double* ms);

#ifdef __cplusplus
}
#endif
#endif /* MAGE_MATCH_H */
