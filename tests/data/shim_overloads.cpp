#include <array>
#include <vector>
#include "BundlerLib.h"
struct Quat { struct C { float v[4]; const float* data() const { return v; } } c; const C& coeffs() const { return c; } };
struct Map3 { float* p; float* data() const { return p; } };      // like Eigen::Map<Vector3f>: data() on a const object gives a mutable pointer
void all_forms(mage::BundlerLib& b)
{
    float t3[3] = {}, r9[9] = {}, k4[4] = {}, uv2[2] = {}, q4[4] = {};
    float* pt = t3; const float* cpt = t3;
    std::array<float, 3> at{}; std::array<float, 9> ar{}; std::array<float, 4> ak{}; std::array<float, 2> auv{};
    std::vector<float> vt(3), vr(9), vk(4), vuv(2), hub(2, 1.8f);
    std::vector<unsigned> out;
    b.SetCameraPose(0, t3, r9, k4, false); b.SetCameraPose(0, pt, r9, cpt, true); b.SetCameraPose(0, at, ar, ak, false); b.SetCameraPose(0, vt, vr, vk, false);
    b.SetCameraPose(0, cpt, ar, vk, false);
    b.SetMapPoint(0, t3); b.SetMapPoint(0, pt); b.SetMapPoint(0, cpt); b.SetMapPoint(0, at); b.SetMapPoint(0, vt);
    b.SetObservation(0, uv2, 0, 0, 1.0f); b.SetObservation(0, auv, 0, 0, 1.0f); b.SetObservation(0, vuv, 0, 0, 1.0f); b.SetObservation(0, cpt, 0, 0, 1.0f);
    b.SetRelativeRotationConstraint(0, 0, 1, q4); b.SetRelativeRotationConstraint(0, 0, 1, Quat{}, 2.0f); b.SetRelativeRotationConstraint(0, 0, 1, cpt);
    b.SetRelativeTransformConstraint(0, 0, 1, t3, q4, 1.0f); b.SetRelativeTransformConstraint(0, 0, 1, at, Quat{}, 1.0f); b.SetRelativeTransformConstraint(0, 0, 1, cpt, cpt, 1.0f);
    b.GetPose(0, t3, r9); b.GetPose(0, pt, r9); b.GetPose(0, at, ar); b.GetPose(0, vt, vr); b.GetPose(0, Map3{ t3 }, Map3{ r9 });
    b.GetPoint(0, t3); b.GetPoint(0, pt); b.GetPoint(0, at); b.GetPoint(0, vt); b.GetPoint(0, Map3{ t3 });
    b.StepBundleAdjustment(hub, 7.25f, out); b.StepBundleAdjustment(hub.data(), hub.size(), 7.25f, out);
}

#include "FeatureMatcher.h"
struct Desc32 { unsigned char b[32]; };
struct DMatchLike { int queryIdx, trainIdx, imgIdx; float distance; };
unsigned indexed_match_forms(mage::MatcherContext& ctx, const mage::BowTree& tree)
{
    std::vector<Desc32> da(4), db(4);
    std::vector<int32_t> fao(tree.childOffsets.size(), 0), fa, fbo(tree.childOffsets.size(), 0), fb;
    std::vector<bool> ma, mb;
    std::vector<DMatchLike> good;
    unsigned n = mage::IndexedMatch(ctx, tree, da, fao, fa, db, fbo, fb, ma, mb, 30, 1, good);      // the tree staged with the call
    mage::SetBowTree(ctx, tree);
    n += mage::IndexedMatchResidentTree(ctx, da, fao, fa, db, fbo, fb, ma, mb, 30, 1, good);         // the tree the context keeps on the device
    return n + static_cast<unsigned>(mage::FindLeafNodes(ctx, da).size() + mage::FindLeafNodes(ctx, tree, db).size());
}
