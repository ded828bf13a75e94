// The reference's local-BA caller loop (BundleAdjust.cpp:281-354: shrinking outlier threshold, one StepBundleAdjustment per
// threshold, `outliers` appended to by the callee) written against include/BundlerLib.h exactly as the reference writes it --
// per-element Set* calls, no capacity hint, no extension call -- on a scene file written by mageslam_amd/scene.py::save_scene.
//   g++ -std=c++17 -Iinclude tools/shim_local_ba.cpp -Lmageslam_amd -lmageslam_hip -Wl,-rpath,$PWD/mageslam_amd -o tools/_bin/shim_local_ba
//   shim_local_ba scene.bin huber thr0 [thr1 ...]      prints: "step <mse> <n_new>" per step, then "outliers i0 i1 ..."
//   shim_local_ba scene.bin huber --steady N            what ONE one-iteration StepBundleAdjustment costs this (C++) caller once the
//                                                       structure is built: 2 untimed calls, then N timed ones at maxErrSq = 1e30;
//                                                       prints "steady_ms <median> <min>" (bench.py extra.config3)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "BundlerLib.h"
#include "scene_io.h"

int main(int argc, char** argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: %s scene.bin huber thr0 [thr1 ...]\n", argv[0]); return 2; }
    try {
        const SceneFile s = read_scene(argv[1]);
        const float huber = (float)std::atof(argv[2]);
        mage::BundlerLib bundler{ mage::BundlerParameters{ false } };
        bundler.AllocateCameras(s.n_cams);
        for (size_t i = 0; i < s.n_cams; ++i) bundler.SetCameraPose(i, &s.cam_t[i * 3], &s.cam_R[i * 9], &s.cam_K[i * 4], s.cam_fixed[i] != 0);
        bundler.AllocateMapPoints(s.n_pts);
        for (size_t i = 0; i < s.n_pts; ++i) bundler.SetMapPoint(i, &s.points[i * 3]);
        bundler.AllocateObservations(s.n_obs);
        for (size_t i = 0; i < s.n_obs; ++i) bundler.SetObservation(i, &s.obs_uv[i * 2], s.obs_cam[i], s.obs_pt[i], s.obs_info[i]);
        std::vector<unsigned int> outliers;                      // the caller's vector: appended to, never cleared by the callee
        if (argc >= 5 && std::strcmp(argv[3], "--steady") == 0) {
            const int n = std::atoi(argv[4]);
            const std::vector<float> widths{ huber };
            std::vector<double> ms;
            for (int i = 0; i < n + 2; ++i) {
                const auto t0 = std::chrono::steady_clock::now();
                (void)bundler.StepBundleAdjustment(widths, 1e30f, outliers);
                if (i >= 2) ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            }
            std::sort(ms.begin(), ms.end());
            std::printf("steady_ms %.5f %.5f\n", ms.empty() ? 0.0 : ms[ms.size() / 2], ms.empty() ? 0.0 : ms[0]);
            return 0;
        }
        for (int a = 3; a < argc; ++a) {
            const size_t before = outliers.size();
            const std::vector<float> widths{ huber };
            const float mse = bundler.StepBundleAdjustment(widths, (float)std::atof(argv[a]), outliers);
            std::printf("step %.9g %zu\n", mse, outliers.size() - before);
        }
        std::printf("outliers");
        for (unsigned int o : outliers) std::printf(" %u", o);
        std::printf("\n");
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
