// FeatureMatcher.h -- C++ shim with the reference's function names and argument order over the MI355X C ABI (mage_match.h).
//
// Replaces Core/MAGESLAM/Source/Tracking/FeatureMatcher.h:29-136 (namespace mage):
//   Match               :66-75     two-way brute-force Hamming match of two analysed images
//   RadiusMatch         :90-103    multi-query form;  :118-130 single-query form
//   IndexedMatch        :29-43     with the candidate lists the vocabulary index returned, or looked up in the vocabulary tree on the device (BowTree)
//   GetDescriptorDistance / GetDescriptorDistanceSlow :132-134
// The reference hands over shared_ptr<AnalyzedImage> (keypoints + descriptors), std::vector<bool> masks, a KeypointSpatialIndex and a
// thread_memory; what the algorithm reads of them is the keypoint and descriptor arrays, so the shim takes those: anything with
// data() / size() whose elements have cv::KeyPoint's 28-byte layout (= mage_keypoint) or are 32-byte descriptors; cv::DMatch's
// layout (queryIdx, trainIdx, imgIdx, distance = mage_dmatch) for the result.  The spatial index is implied (the kernel scans the
// ~440 targets with a box test), the temporary memory is the handle's.  Results are APPENDED to goodMatches, as the reference does,
// and the number of new matches is returned.
//
// A process needs one matcher handle per thread that matches: MatcherContext owns it; the overloads without a context use a
// thread_local one.  Errors throw std::runtime_error carrying mage_last_error().
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "mage_match.h"

namespace mage
{
    class MatcherContext
    {
    public:
        explicit MatcherContext(int device = -1)
        {
            mage_matcher* h = nullptr;
            if (mage_matcher_create(device, &h) != MAGE_OK) throw std::runtime_error(std::string("MatcherContext: ") + mage_last_error());
            m_impl.reset(h);
        }
        mage_matcher* Handle() const { return m_impl.get(); }
        static MatcherContext& ThreadDefault() { thread_local MatcherContext ctx; return ctx; }

    private:
        struct Deleter { void operator()(mage_matcher* h) const { mage_matcher_destroy(h); } };
        std::unique_ptr<mage_matcher, Deleter> m_impl;
    };

    namespace shim
    {
        inline void CheckMatch(mage_status s, const char* who)
        {
            if (s != MAGE_OK) throw std::runtime_error(std::string(who) + ": " + mage_last_error());
        }
        // std::vector<bool> (the reference's mask type) -> one byte per element; null = all
        inline const uint8_t* Bytes(const std::vector<bool>* mask, std::vector<uint8_t>& store)
        {
            if (!mask) return nullptr;
            store.resize(mask->size());
            for (size_t i = 0; i < mask->size(); ++i) store[i] = (*mask)[i] ? 1 : 0;
            return store.data();
        }
        template <typename DMatchVec, typename Call>
        unsigned int Append(DMatchVec& goodMatches, size_t most, Call&& call)
        {
            static_assert(sizeof(typename DMatchVec::value_type) == sizeof(mage_dmatch), "matches must have cv::DMatch's 16-byte layout");
            const size_t old = goodMatches.size();
            goodMatches.resize(old + most);
            int count = 0;
            call(reinterpret_cast<mage_dmatch*>(goodMatches.data() + old), static_cast<int>(most), &count);
            goodMatches.resize(old + static_cast<size_t>(count));
            return static_cast<unsigned int>(count);
        }
    }

    // Globally matches descriptors based on their Hamming distance (FeatureMatcher.cpp:61-190).  Indices in the result are indices
    // into descriptorsA / descriptorsB; masks select the descriptors that take part (the reference also passes their popcounts).
    template <typename DescriptorsA, typename DescriptorsB, typename DMatchVec>
    unsigned int Match(MatcherContext& ctx, const DescriptorsA& descriptorsA, const DescriptorsB& descriptorsB, const std::vector<bool>& imageAMask,
                       const std::vector<bool>& imageBMask, int maxHammingDist, int minHammingDifference, DMatchVec& goodMatches)
    {
        static_assert(sizeof(*descriptorsA.data()) == 32 && sizeof(*descriptorsB.data()) == 32, "descriptors are 32 bytes");
        std::vector<uint8_t> ma, mb;
        const uint8_t* pa = shim::Bytes(imageAMask.empty() ? nullptr : &imageAMask, ma);
        const uint8_t* pb = shim::Bytes(imageBMask.empty() ? nullptr : &imageBMask, mb);
        const int nA = static_cast<int>(descriptorsA.size()), nB = static_cast<int>(descriptorsB.size());
        return shim::Append(goodMatches, static_cast<size_t>(nA), [&](mage_dmatch* out, int cap, int* count) {
            shim::CheckMatch(mage_match_masked(ctx.Handle(), reinterpret_cast<const uint8_t*>(descriptorsA.data()), nA, pa,
                                               reinterpret_cast<const uint8_t*>(descriptorsB.data()), nB, pb, maxHammingDist, minHammingDifference, out, cap, count), "Match");
        });
    }
    template <typename DescriptorsA, typename DescriptorsB, typename DMatchVec>
    unsigned int Match(const DescriptorsA& descriptorsA, const DescriptorsB& descriptorsB, const std::vector<bool>& imageAMask, const std::vector<bool>& imageBMask,
                       int maxHammingDist, int minHammingDifference, DMatchVec& goodMatches)
    {
        return Match(MatcherContext::ThreadDefault(), descriptorsA, descriptorsB, imageAMask, imageBMask, maxHammingDist, minHammingDifference, goodMatches);
    }

    // RadiusMatch, multi-query form (FeatureMatcher.cpp:294-378).  queryKeypointPositionOverrides: any vector of 8-byte (x, y) float
    // records (cv::Point2f) or null; masks null = all.
    template <typename QueryKeypoints, typename Point2fVec, typename QueryDescriptors, typename TargetKeypoints, typename TargetDescriptors, typename DMatchVec>
    unsigned int RadiusMatch(MatcherContext& ctx, const QueryKeypoints& queryKeypoints, const Point2fVec* queryKeypointPositionOverrides,
                             const std::vector<bool>* queryKeypointsMask, const QueryDescriptors& queryDescriptors, const TargetKeypoints& targetKeypoints,
                             const std::vector<bool>* targetKeypointsMask, const TargetDescriptors& targetDescriptors, float radius, int maxHammingDist,
                             int minHammingDifference, DMatchVec& goodMatches)
    {
        static_assert(sizeof(*queryKeypoints.data()) == sizeof(mage_keypoint) && sizeof(*targetKeypoints.data()) == sizeof(mage_keypoint), "cv::KeyPoint layout");
        static_assert(sizeof(*queryDescriptors.data()) == 32 && sizeof(*targetDescriptors.data()) == 32, "descriptors are 32 bytes");
        static_assert(sizeof(typename Point2fVec::value_type) == 8, "position overrides are (x, y) float pairs");
        std::vector<uint8_t> mq, mt;
        const uint8_t* pq = shim::Bytes(queryKeypointsMask, mq);
        const uint8_t* pt = shim::Bytes(targetKeypointsMask, mt);
        const int nQ = static_cast<int>(queryKeypoints.size()), nT = static_cast<int>(targetKeypoints.size());
        return shim::Append(goodMatches, static_cast<size_t>(nQ), [&](mage_dmatch* out, int cap, int* count) {
            shim::CheckMatch(mage_match_radius(ctx.Handle(), reinterpret_cast<const mage_keypoint*>(queryKeypoints.data()), nQ,
                                               queryKeypointPositionOverrides ? reinterpret_cast<const float*>(queryKeypointPositionOverrides->data()) : nullptr, pq,
                                               reinterpret_cast<const uint8_t*>(queryDescriptors.data()), reinterpret_cast<const mage_keypoint*>(targetKeypoints.data()), nT,
                                               pt, reinterpret_cast<const uint8_t*>(targetDescriptors.data()), radius, maxHammingDist, minHammingDifference, out, cap, count),
                             "RadiusMatch");
        });
    }

    // RadiusMatch, single-query form (FeatureMatcher.cpp:386-446): true and bestMatch filled when the query found a match.
    template <typename KeyPoint, typename Descriptor, typename TargetKeypoints, typename TargetDescriptors, typename DMatch>
    bool RadiusMatch(MatcherContext& ctx, const KeyPoint& queryKeypoint, const float* queryKeypointPositionOverrideXY, const Descriptor& queryDescriptor,
                     const TargetKeypoints& targetKeypoints, const std::vector<bool>* targetKeypointsMask, const TargetDescriptors& targetDescriptors, float radius,
                     int maxHammingDist, int minHammingDifference, DMatch& bestMatch)
    {
        static_assert(sizeof(KeyPoint) == sizeof(mage_keypoint) && sizeof(Descriptor) == 32 && sizeof(DMatch) == sizeof(mage_dmatch), "record layouts");
        std::vector<uint8_t> mt;
        const uint8_t* pt = shim::Bytes(targetKeypointsMask, mt);
        mage_dmatch m{};
        int count = 0;
        shim::CheckMatch(mage_match_radius(ctx.Handle(), reinterpret_cast<const mage_keypoint*>(&queryKeypoint), 1, queryKeypointPositionOverrideXY, nullptr,
                                           reinterpret_cast<const uint8_t*>(&queryDescriptor), reinterpret_cast<const mage_keypoint*>(targetKeypoints.data()),
                                           static_cast<int>(targetKeypoints.size()), pt, reinterpret_cast<const uint8_t*>(targetDescriptors.data()), radius,
                                           maxHammingDist, minHammingDifference, &m, 1, &count), "RadiusMatch");
        if (count < 1) return false;
        bestMatch = *reinterpret_cast<const DMatch*>(&m);
        return true;
    }

    // IndexedMatch (FeatureMatcher.cpp:192-292) with the candidate lists of BaseBow::QueryFeatures as CSR arrays (offsets[n + 1], items)
    template <typename DescriptorsA, typename DescriptorsB, typename DMatchVec>
    unsigned int IndexedMatch(MatcherContext& ctx, const DescriptorsA& descriptorsA, const std::vector<int32_t>& candidatesInBOffsets, const std::vector<int32_t>& candidatesInB,
                              const DescriptorsB& descriptorsB, const std::vector<int32_t>& candidatesInAOffsets, const std::vector<int32_t>& candidatesInA,
                              const std::vector<bool>& imageAMask, const std::vector<bool>& imageBMask, int maxHammingDist, int minHammingDifference, DMatchVec& goodMatches)
    {
        std::vector<uint8_t> ma, mb;
        const uint8_t* pa = shim::Bytes(imageAMask.empty() ? nullptr : &imageAMask, ma);
        const uint8_t* pb = shim::Bytes(imageBMask.empty() ? nullptr : &imageBMask, mb);
        const int nA = static_cast<int>(descriptorsA.size()), nB = static_cast<int>(descriptorsB.size());
        return shim::Append(goodMatches, static_cast<size_t>(nA), [&](mage_dmatch* out, int cap, int* count) {
            shim::CheckMatch(mage_match_indexed(ctx.Handle(), reinterpret_cast<const uint8_t*>(descriptorsA.data()), nA, pa, candidatesInBOffsets.data(), candidatesInB.data(),
                                                reinterpret_cast<const uint8_t*>(descriptorsB.data()), nB, pb, candidatesInAOffsets.data(), candidatesInA.data(),
                                                maxHammingDist, minHammingDifference, out, cap, count), "IndexedMatch");
        });
    }

    // The vocabulary tree of BoW/OnlineBow as the flat arrays of mage_bow_tree (node medoids, child lists in childrenIDs order; node 0 = root)
    struct BowTree
    {
        std::vector<uint8_t> nodeDescriptors;      // 32 bytes per node
        std::vector<int32_t> childOffsets, children;
        mage_bow_tree View() const { return mage_bow_tree{ nodeDescriptors.data(), childOffsets.data(), children.data(), static_cast<int32_t>(childOffsets.size()) - 1 }; }
    };
    // Keeps the (trained) tree on the device: FindLeafNodes(ctx, descriptors) then moves only the descriptors (mage_bow_set_tree)
    inline void SetBowTree(MatcherContext& ctx, const BowTree& tree)
    {
        const mage_bow_tree t = tree.View();
        shim::CheckMatch(mage_bow_set_tree(ctx.Handle(), &t), "SetBowTree");
    }
    template <typename Descriptors>
    std::vector<int32_t> FindLeafNodes(MatcherContext& ctx, const Descriptors& descriptors)
    {
        std::vector<int32_t> leaves(descriptors.size());
        shim::CheckMatch(mage_bow_find_leaf_batch(ctx.Handle(), nullptr, reinterpret_cast<const uint8_t*>(descriptors.data()), static_cast<int>(descriptors.size()), leaves.data()), "FindLeafNodes");
        return leaves;
    }
    // OnlineBow::FindLeafNode (BoW/OnlineBow.cpp:289-311) for a batch of descriptors
    template <typename Descriptors>
    std::vector<int32_t> FindLeafNodes(MatcherContext& ctx, const BowTree& tree, const Descriptors& descriptors)
    {
        std::vector<int32_t> leaves(descriptors.size());
        const mage_bow_tree t = tree.View();
        shim::CheckMatch(mage_bow_find_leaf_batch(ctx.Handle(), &t, reinterpret_cast<const uint8_t*>(descriptors.data()), static_cast<int>(descriptors.size()), leaves.data()), "FindLeafNodes");
        return leaves;
    }
    // IndexedMatch (FeatureMatcher.cpp:192-292) as the reference runs it, candidate lists from the vocabulary (BaseBow::QueryFeatures): per node the
    // features of image A / image B filed under it (m_NodeKeyframeMap[node][keyframe].indexes), CSR over the tree's nodes
    template <typename DescriptorsA, typename DescriptorsB, typename DMatchVec>
    unsigned int IndexedMatch(MatcherContext& ctx, const BowTree& tree, const DescriptorsA& descriptorsA, const std::vector<int32_t>& leafFeaturesAOffsets, const std::vector<int32_t>& leafFeaturesA,
                              const DescriptorsB& descriptorsB, const std::vector<int32_t>& leafFeaturesBOffsets, const std::vector<int32_t>& leafFeaturesB,
                              const std::vector<bool>& imageAMask, const std::vector<bool>& imageBMask, int maxHammingDist, int minHammingDifference, DMatchVec& goodMatches)
    {
        std::vector<uint8_t> ma, mb;
        const uint8_t* pa = shim::Bytes(imageAMask.empty() ? nullptr : &imageAMask, ma);
        const uint8_t* pb = shim::Bytes(imageBMask.empty() ? nullptr : &imageBMask, mb);
        const int nA = static_cast<int>(descriptorsA.size()), nB = static_cast<int>(descriptorsB.size());
        const mage_bow_tree t = tree.View();
        return shim::Append(goodMatches, static_cast<size_t>(nA), [&](mage_dmatch* out, int cap, int* count) {
            shim::CheckMatch(mage_match_indexed_bow(ctx.Handle(), &t, reinterpret_cast<const uint8_t*>(descriptorsA.data()), nA, pa, leafFeaturesAOffsets.data(), leafFeaturesA.data(),
                                                    reinterpret_cast<const uint8_t*>(descriptorsB.data()), nB, pb, leafFeaturesBOffsets.data(), leafFeaturesB.data(),
                                                    maxHammingDist, minHammingDifference, out, cap, count), "IndexedMatch");
        });
    }

    // the same through the tree the context keeps on the device (SetBowTree): only the frame's descriptors and lists travel.  (A name of its
    // own: without the tree the argument list is that of the IndexedMatch over caller-built candidate lists above.)
    template <typename DescriptorsA, typename DescriptorsB, typename DMatchVec>
    unsigned int IndexedMatchResidentTree(MatcherContext& ctx, const DescriptorsA& descriptorsA, const std::vector<int32_t>& leafFeaturesAOffsets, const std::vector<int32_t>& leafFeaturesA,
                              const DescriptorsB& descriptorsB, const std::vector<int32_t>& leafFeaturesBOffsets, const std::vector<int32_t>& leafFeaturesB,
                              const std::vector<bool>& imageAMask, const std::vector<bool>& imageBMask, int maxHammingDist, int minHammingDifference, DMatchVec& goodMatches)
    {
        std::vector<uint8_t> ma, mb;
        const uint8_t* pa = shim::Bytes(imageAMask.empty() ? nullptr : &imageAMask, ma);
        const uint8_t* pb = shim::Bytes(imageBMask.empty() ? nullptr : &imageBMask, mb);
        const int nA = static_cast<int>(descriptorsA.size()), nB = static_cast<int>(descriptorsB.size());
        return shim::Append(goodMatches, static_cast<size_t>(nA), [&](mage_dmatch* out, int cap, int* count) {
            shim::CheckMatch(mage_match_indexed_bow(ctx.Handle(), nullptr, reinterpret_cast<const uint8_t*>(descriptorsA.data()), nA, pa, leafFeaturesAOffsets.data(), leafFeaturesA.data(),
                                                    reinterpret_cast<const uint8_t*>(descriptorsB.data()), nB, pb, leafFeaturesBOffsets.data(), leafFeaturesB.data(),
                                                    maxHammingDist, minHammingDifference, out, cap, count), "IndexedMatch");
        });
    }

    template <typename Descriptor>
    int GetDescriptorDistance(const Descriptor& d0, const Descriptor& d1)
    {
        static_assert(sizeof(Descriptor) == 32, "descriptors are 32 bytes");
        return mage_hamming256(reinterpret_cast<const uint8_t*>(&d0), reinterpret_cast<const uint8_t*>(&d1));
    }
    template <typename Descriptor>
    int GetDescriptorDistanceSlow(const Descriptor& d0, const Descriptor& d1) { return GetDescriptorDistance(d0, d1); }
}
