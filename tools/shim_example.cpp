// Compile check + usage example of the C++ shim (include/BundlerLib.h):
//   g++ -std=c++17 -Iinclude tools/shim_example.cpp -Lmageslam_amd -lmageslam_hip -Wl,-rpath,$PWD/mageslam_amd -o tools/_bin/shim_example
// Mirrors the call order of BundleAdjust.cpp:25-193 / :281-354 on a three-camera toy problem.
#include <array>
#include <cstdio>
#include <vector>

#include "BundlerLib.h"

int main()
{
    try {
        mage::BundlerLib bundler{ mage::BundlerParameters{ false } };
        const std::array<float, 9> R{ 1, 0, 0, 0, 1, 0, 0, 0, 1 };
        const std::array<float, 4> K{ 320, 240, 500, 500 };
        bundler.AllocateCameras(3);
        for (size_t i = 0; i < 3; ++i) bundler.SetCameraPose(i, std::array<float, 3>{ -0.2f * i, 0, 0 }, R, K, i < 2);
        bundler.AllocateMapPoints(4);
        const float P[4][3] = { { 0, 0, 5 }, { 1, 0.5f, 6 }, { -1, 0.3f, 4 }, { 0.2f, -0.7f, 7 } };
        for (size_t p = 0; p < 4; ++p) bundler.SetMapPoint(p, P[p]);
        bundler.AllocateObservations(12);
        size_t o = 0;
        for (size_t p = 0; p < 4; ++p)
            for (size_t c = 0; c < 3; ++c) {
                const float x = P[p][0] - 0.2f * c, z = P[p][2];
                const std::array<float, 2> uv{ 500 * x / z + 320 + 0.3f * (float)c, 500 * P[p][1] / z + 240 };
                bundler.SetObservation(o++, uv, c, p, 0.9f);
            }
        // tethers as a stereo rig hands them over (BundleAdjust.cpp:155-192): a fixed baseline and a relative rotation
        struct Quat { std::array<float, 4> c; const std::array<float, 4>& coeffs() const { return c; } };   // stands in for Eigen::Quaternionf
        bundler.AllocateFixedDistanceConstraints(1);
        bundler.SetFixedDistanceConstraint(0, 1, 2, 0.2f, 10.0f);
        bundler.AllocateRelativeRotationConstraints(1);
        bundler.SetRelativeRotationConstraint(0, 1, 2, Quat{ { 0, 0, 0, 1 } }, 10.0f);
        bundler.AllocateRelativeTransformConstraints(0);
        std::vector<unsigned int> outliers;
        const std::vector<float> huber(3, 1.8f);
        const float mse = bundler.StepBundleAdjustment(huber, 7.25f, outliers);
        std::array<float, 3> t; std::array<float, 9> Rout;
        bundler.GetPose(2, t, Rout);
        std::printf("mse %.6f  outliers %zu  lambda %g  t2 = (%f %f %f)\n", mse, outliers.size(), bundler.GetCurrentLambda(), t[0], t[1], t[2]);
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
