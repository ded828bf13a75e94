// A MAGE-SLAM front-end caller written against include/OrbDetector.h and include/FeatureMatcher.h only (the reference's class
// and function names over the C ABI), with stand-ins for cv::KeyPoint / cv::DMatch / cv::Point2f / ORBDescriptor that have
// their layouts.  Built by tests/test_orb_gpu.py with the host compiler alone and compared with the Python mirror record for record.
//   shim_orb_match a.raw b.raw width height  ->  text on stdout
#include <array>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "FeatureMatcher.h"
#include "OrbDetector.h"

struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };     // cv::KeyPoint
struct DMatch { int queryIdx, trainIdx, imgIdx; float distance; };                     // cv::DMatch
using ORBDescriptor = std::array<uint8_t, 32>;
struct Mat { unsigned char* data; int cols, rows; size_t step; };                        // the members of cv::Mat the detector reads

static std::vector<unsigned char> read_file(const char* path, size_t n)
{
    std::vector<unsigned char> v(n);
    FILE* f = std::fopen(path, "rb");
    if (!f || std::fread(v.data(), 1, n, f) != n) { std::fprintf(stderr, "cannot read %s\n", path); std::exit(2); }
    std::fclose(f);
    return v;
}

int main(int argc, char** argv)
{
    if (argc < 5) return 2;
    const int w = std::atoi(argv[3]), h = std::atoi(argv[4]);
    std::vector<unsigned char> ia = read_file(argv[1], (size_t)w * h), ib = read_file(argv[2], (size_t)w * h);
    try {
        mage::FeatureExtractorSettings settings;                          // the reference's defaults
        mage::OrbFeatureDetector detector(settings);
        std::vector<KeyPoint> ka, kb;
        std::vector<ORBDescriptor> da, db;
        detector.Detector().DetectAndCompute(Mat{ ia.data(), w, h, (size_t)w }, ka, da);
        // Process = DetectAndCompute + UndistortKeypoints
        const float K[9] = { 500.f, 0.f, w / 2.f, 0.f, 500.f, h / 2.f, 0.f, 0.f, 1.f }, Knew[9] = { 480.f, 0.f, w / 2.f, 0.f, 480.f, h / 2.f, 0.f, 0.f, 1.f };
        const float dist[5] = { 0.08f, -0.02f, 0.001f, -0.0005f, 0.004f };
        detector.Process(K, dist, 5, Knew, ib.data(), w, h, w, kb, db);
        std::printf("A %zu B %zu\n", ka.size(), kb.size());
        for (size_t i = 0; i < ka.size(); ++i) std::printf("ka %g %g %g %d\n", ka[i].pt.x, ka[i].pt.y, ka[i].response, (int)da[i][0] | ((int)da[i][31] << 8));
        for (size_t i = 0; i < kb.size(); ++i) std::printf("kb %.9g %.9g %g %d\n", kb[i].pt.x, kb[i].pt.y, kb[i].response, (int)db[i][0] | ((int)db[i][31] << 8));
        std::vector<DMatch> good;
        std::vector<bool> maskA(ka.size(), true), maskB;                   // an empty mask = all
        for (size_t i = 0; i < maskA.size(); i += 5) maskA[i] = false;
        const unsigned n1 = mage::Match(da, db, maskA, maskB, 40, 2, good);
        std::printf("match %u\n", n1);
        for (const DMatch& m : good) std::printf("m %d %d %d %g\n", m.queryIdx, m.trainIdx, m.imgIdx, m.distance);
        mage::MatcherContext ctx;
        std::vector<DMatch> rad;
        std::vector<Point2f> overrides(ka.size());
        for (size_t i = 0; i < ka.size(); ++i) overrides[i] = Point2f{ ka[i].pt.x + 1.5f, ka[i].pt.y - 0.5f };
        const unsigned n2 = mage::RadiusMatch(ctx, ka, &overrides, static_cast<const std::vector<bool>*>(nullptr), da, kb, static_cast<const std::vector<bool>*>(nullptr), db, 12.0f, 50, 1, rad);
        std::printf("radius %u\n", n2);
        for (const DMatch& m : rad) std::printf("r %d %d %d %g\n", m.queryIdx, m.trainIdx, m.imgIdx, m.distance);
        int singles = 0;
        for (size_t i = 0; i < ka.size() && i < 20; ++i) {
            DMatch best{};
            if (mage::RadiusMatch(ctx, ka[i], static_cast<const float*>(nullptr), da[i], kb, static_cast<const std::vector<bool>*>(nullptr), db, 12.0f, 50, 1, best)) {
                ++singles;
                std::printf("s %zu %d %g\n", i, best.trainIdx, best.distance);
            }
        }
        std::printf("singles %d\n", singles);
        if (!da.empty() && !db.empty()) std::printf("dist %d %d\n", mage::GetDescriptorDistance(da[0], db[0]), mage::GetDescriptorDistanceSlow(da[0], da[0]));
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
