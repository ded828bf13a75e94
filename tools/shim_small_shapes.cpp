// The reference's two everyday callers of BundlerLib, written against include/BundlerLib.h the way the reference writes them
// (a bundler per optimisation, per-element Set* / Get* calls, no extension call), timed phase by phase:
//
//   shim_small_shapes pose-only scene.bin [frames]
//       TrackLocalMap::OptimizeCameraPose (Tracking/TrackLocalMap.cpp:421-501), called twice per frame (:94-105 and the second pass
//       after the covisibility search): make_unique<BundlerLib>(ArePointsFixed) -> AllocateCameras(1) / SetCameraPose ->
//       AllocateMapPoints / AllocateObservations, SetMapPoint + SetObservation per point -> ONE StepBundleAdjustment of
//       InitialPoseEstimateBundleAdjustmentSteps = 3 iterations at Huber 4.0 and MaxOutlierErrorPoseEstimation^2 = 36 (pass 1), of
//       BundleAdjustmentG2OSteps = 4 iterations at Huber 0.9 and MaxOutlierError^2 = 20.25 (pass 2; MageSettings.h:180-195) ->
//       GetPose(0) -> reset().  The scene file holds one camera and its map points.
//   shim_small_shapes window scene.bin [runs]
//       BundleAdjust::RunBundleAdjustment with the default BundleAdjustSettings (MageSettings.h:41-52: NumSteps = NumStepsPerRun = 1,
//       Huber 1.8, MaxOutlierError 7.25 -- passed un-squared, BundleAdjust.cpp:303): MakeBundler -> BuildDataForG2O (per-element
//       setters, BundleAdjust.cpp:25-193) -> one StepBundleAdjustment({1.8}, 7.25) -> UpdateData (GetPose per free keyframe, GetPoint
//       per map point, :195-222) -> reset().
// Output: one line "pose_only_ms ..." / "window_ms ..." of medians in milliseconds (bench.py extra.pose_only / extra.reference_window).
//   g++ -O2 -std=c++17 -Iinclude tools/shim_small_shapes.cpp -Lmageslam_amd -lmageslam_hip -Wl,-rpath,$PWD/mageslam_amd -o tools/_bin/shim_small_shapes
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "BundlerLib.h"
#include "scene_io.h"

namespace {
using Clock = std::chrono::steady_clock;
double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
double median(std::vector<double> v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
double minimum(const std::vector<double>& v) { return v.empty() ? 0.0 : *std::min_element(v.begin(), v.end()); }

struct Phases { std::vector<double> create, set, step, get, destroy, total; };

// TrackLocalMap::OptimizeCameraPose, one call
void optimize_camera_pose(const SceneFile& s, unsigned iterations, float huber, float max_err_sq, Phases& ph, float pos[3], float rot[9], size_t* n_out)
{
    const auto t0 = Clock::now();
    mage::BundlerParameters params{};
    params.ArePointsFixed = true;
    auto ba = std::make_unique<mage::BundlerLib>(params);
    const auto t1 = Clock::now();
    ba->AllocateCameras(1);
    ba->SetCameraPose(0, &s.cam_t[0], &s.cam_R[0], &s.cam_K[0], /*isFixed*/ false);
    ba->AllocateMapPoints(s.n_obs);
    ba->AllocateObservations(s.n_obs);
    for (uint32_t i = 0; i < s.n_obs; ++i) {
        ba->SetMapPoint(i, &s.points[(size_t)s.obs_pt[i] * 3]);
        ba->SetObservation(i, &s.obs_uv[(size_t)i * 2], 0, i, s.obs_info[i]);
    }
    const auto t2 = Clock::now();
    std::vector<unsigned int> outliers;
    outliers.reserve(s.n_obs);
    const std::vector<float> widths(iterations, huber);
    ba->StepBundleAdjustment(widths, max_err_sq, outliers);
    const auto t3 = Clock::now();
    ba->GetPose(0, &pos[0], &rot[0]);
    const auto t4 = Clock::now();
    ba.reset();
    const auto t5 = Clock::now();
    auto d = [](Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    ph.create.push_back(d(t0, t1)); ph.set.push_back(d(t1, t2)); ph.step.push_back(d(t2, t3)); ph.get.push_back(d(t3, t4));
    ph.destroy.push_back(d(t4, t5)); ph.total.push_back(d(t0, t5));
    if (n_out) *n_out = outliers.size();
}

// BundleAdjust::RunBundleAdjustment with NumSteps = 1
void run_bundle_adjustment(const SceneFile& s, Phases& ph, float* mse_out, size_t* n_out, std::vector<float>& sink)
{
    const auto t0 = Clock::now();
    mage::BundlerParameters params{};
    auto bundler = std::make_unique<mage::BundlerLib>(params);
    const auto t1 = Clock::now();
    bundler->AllocateCameras(s.n_cams);
    for (size_t i = 0; i < s.n_cams; ++i) bundler->SetCameraPose(i, &s.cam_t[i * 3], &s.cam_R[i * 9], &s.cam_K[i * 4], s.cam_fixed[i] != 0);
    bundler->AllocateMapPoints(s.n_pts);
    for (size_t i = 0; i < s.n_pts; ++i) bundler->SetMapPoint(i, &s.points[i * 3]);
    bundler->AllocateObservations(s.n_obs);
    for (size_t i = 0; i < s.n_obs; ++i) bundler->SetObservation(i, &s.obs_uv[i * 2], s.obs_cam[i], s.obs_pt[i], s.obs_info[i]);
    const auto t2 = Clock::now();
    std::vector<unsigned int> outliers;
    const std::vector<float> widths(1, 1.8f);
    const float mse = bundler->StepBundleAdjustment(widths, 7.25f, outliers);
    const auto t3 = Clock::now();
    float pos[3], rot[9], p[3];
    for (size_t i = 0; i < s.n_cams; ++i)
        if (!s.cam_fixed[i]) { bundler->GetPose(i, &pos[0], &rot[0]); sink[0] += pos[0] + rot[0]; }
    for (size_t i = 0; i < s.n_pts; ++i) { bundler->GetPoint(i, &p[0]); sink[0] += p[0]; }
    const auto t4 = Clock::now();
    bundler.reset();
    const auto t5 = Clock::now();
    auto d = [](Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    ph.create.push_back(d(t0, t1)); ph.set.push_back(d(t1, t2)); ph.step.push_back(d(t2, t3)); ph.get.push_back(d(t3, t4));
    ph.destroy.push_back(d(t4, t5)); ph.total.push_back(d(t0, t5));
    if (mse_out) *mse_out = mse;
    if (n_out) *n_out = outliers.size();
}

void print_phases(const char* tag, const Phases& ph)
{
    std::printf("%s total %.5f min %.5f create %.5f set %.5f step %.5f get %.5f destroy %.5f calls %zu\n", tag, median(ph.total), minimum(ph.total),
                median(ph.create), median(ph.set), median(ph.step), median(ph.get), median(ph.destroy), ph.total.size());
}
}  // namespace

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s pose-only|window scene.bin [repetitions]\n", argv[0]); return 2; }
    try {
        const SceneFile s = read_scene(argv[2]);
        const int reps = argc > 3 ? std::atoi(argv[3]) : 200;
        if (std::strcmp(argv[1], "pose-only") == 0) {
            if (s.n_cams != 1) { std::fprintf(stderr, "pose-only wants a scene of one camera\n"); return 2; }
            Phases warm, p1, p2;
            float pos[3], rot[9];
            size_t n1 = 0, n2 = 0;
            for (int i = 0; i < 5; ++i) { optimize_camera_pose(s, 3, 4.0f, 36.0f, warm, pos, rot, nullptr); optimize_camera_pose(s, 4, 0.9f, 20.25f, warm, pos, rot, nullptr); }
            const auto t0 = Clock::now();
            for (int i = 0; i < reps; ++i) {
                optimize_camera_pose(s, 3, 4.0f, 36.0f, p1, pos, rot, &n1);          // pass 1 of the frame
                optimize_camera_pose(s, 4, 0.9f, 20.25f, p2, pos, rot, &n2);         // pass 2
            }
            const double per_frame = ms_since(t0) / reps;
            print_phases("pose_only_pass1_ms", p1);
            print_phases("pose_only_pass2_ms", p2);
            std::printf("pose_only_frame_ms %.5f outliers %zu %zu position %.6f %.6f %.6f\n", per_frame, n1, n2, pos[0], pos[1], pos[2]);
        } else if (std::strcmp(argv[1], "window") == 0) {
            Phases warm, ph;
            float mse = 0;
            size_t nout = 0;
            std::vector<float> sink(1, 0.f);
            for (int i = 0; i < 5; ++i) run_bundle_adjustment(s, warm, nullptr, nullptr, sink);
            sink[0] = 0.f;
            for (int i = 0; i < reps; ++i) run_bundle_adjustment(s, ph, &mse, &nout, sink);
            print_phases("window_ms", ph);
            std::printf("window_result mse %.6f outliers %zu checksum %.4f\n", mse, nout, sink[0] / reps);
        } else { std::fprintf(stderr, "unknown mode %s\n", argv[1]); return 2; }
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
