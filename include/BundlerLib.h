// BundlerLib.h -- C++ shim with the reference's class name and method names over the MI355X C ABI (mage_ba.h).
//
// Drop-in for Dependencies/BundlerLib/Include/BundlerLib.h:15-66: MAGE-SLAM's callers
// (Core/MAGESLAM/Source/BundleAdjustment/BundleAdjust.cpp:25-236, Tracking/TrackLocalMap.cpp:439-494) compile against
// this header unchanged and link libmageslam_hip.so instead of BundlerLib + g2o.  The reference passes
// Eigen::Map<> / gsl::span<> views; those types are templates over raw pointers, so the shim takes anything that
// exposes .data() (Eigen::Map, gsl::span, std::array, std::vector) plus .size() for spans -- when Eigen / GSL are on the
// include path the original call sites bind to these overloads as written; without them the raw-pointer overloads work.
//
// Error behaviour: the reference asserts / throws gsl::narrowing_error; the shim throws std::runtime_error carrying
// mage_last_error() when the C ABI reports a failure.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "mage_ba.h"

namespace mage
{
    struct BundlerParameters
    {
        bool ArePointsFixed{ false };        // True if map points should not be optimized
    };

    // What a setter / getter accepts for a vector, matrix or quaternion argument: anything with data() (Eigen::Map, std::array, std::vector,
    // gsl::span) -- the reference's own Eigen::Map callers (Dependencies/BundlerLib/Include/BundlerLib.h:28-58) -- or a raw float pointer /
    // array (a C-style integrator).  One template per method over these helpers: a raw array can no longer deduce a `.data()` overload.
    namespace detail
    {
        template <typename T> inline auto cdata(const T& v) -> decltype(static_cast<const float*>(v.data())) { return v.data(); }
        inline const float* cdata(const float* p) { return p; }
        template <typename T> inline auto mdata(T&& v) -> decltype(static_cast<float*>(v.data())) { return v.data(); }
        inline float* mdata(float* p) { return p; }
        template <typename Q> inline auto qdata(const Q& q) -> decltype(static_cast<const float*>(q.coeffs().data())) { return q.coeffs().data(); }      // x, y, z, w (Eigen::Quaternionf)
        inline const float* qdata(const float* xyzw) { return xyzw; }
    }

    class BundlerLib
    {
    public:
        explicit BundlerLib(const BundlerParameters& bundlerParameters, int device = -1)
            : m_bundlerParameters(bundlerParameters)
        {
            mage_ba_params p{ bundlerParameters.ArePointsFixed ? 1 : 0, device };
            mage_ba* h = nullptr;
            Check(mage_ba_create(&p, &h));
            m_impl.reset(h);
        }
        ~BundlerLib() = default;
        BundlerLib(const BundlerLib&) = delete;
        BundlerLib& operator=(const BundlerLib&) = delete;

        void AllocateCameras(size_t count) { Check(mage_ba_alloc_cameras(m_impl.get(), count)); }

        // position: 3 floats; orientation: 3x3 column-major rotation (world -> camera); intrinsics: cx, cy, fx, fy
        template <typename V3, typename M3, typename V4>
        void SetCameraPose(size_t idx, const V3& position, const M3& orientation, const V4& intrinsics, bool isFixed)
        {
            Check(mage_ba_set_camera(m_impl.get(), idx, detail::cdata(position), detail::cdata(orientation), detail::cdata(intrinsics), isFixed ? 1 : 0));
        }

        void FixCameraPose(size_t idx, bool value) { Check(mage_ba_fix_camera(m_impl.get(), idx, value ? 1 : 0)); }

        // EXTENSION (not in the reference class): new poses for cameras already in the graph, no structure rebuild -- what a
        // window of a keyframe-sharded map does to its halo once per outer iteration (mage_ba.h, mageslam_amd/windowed.py).
        void UpdateCameraPoses(size_t count, const uint32_t* indices, const float* positions3, const float* orientations_colmajor9)
        {
            Check(mage_ba_update_camera_poses(m_impl.get(), count, indices, positions3, orientations_colmajor9));
        }

        void AllocateMapPoints(size_t count) { Check(mage_ba_alloc_points(m_impl.get(), count)); }
        template <typename V3>
        void SetMapPoint(size_t idx, const V3& point) { Check(mage_ba_set_point(m_impl.get(), idx, detail::cdata(point))); }

        void AllocateObservations(size_t count) { Check(mage_ba_alloc_observations(m_impl.get(), count)); }
        template <typename V2>
        void SetObservation(size_t idx, const V2& position, size_t cameraIndex, size_t mapPointIndex, float informationMatrixScalar)
        {
            Check(mage_ba_set_observation(m_impl.get(), idx, detail::cdata(position), cameraIndex, mapPointIndex, informationMatrixScalar));
        }

        void AllocateFixedDistanceConstraints(size_t count) { Check(mage_ba_alloc_fixed_distance_constraints(m_impl.get(), count)); }
        void AllocateRelativeRotationConstraints(size_t count) { Check(mage_ba_alloc_relative_rotation_constraints(m_impl.get(), count)); }
        void AllocateRelativeTransformConstraints(size_t count) { Check(mage_ba_alloc_relative_transform_constraints(m_impl.get(), count)); }

        // Tether edges (BundlerLib.h:40-47).  Q is anything with coeffs().data() in x, y, z, w order (Eigen::Quaternionf) or the four
        // coefficients as a float pointer / array.
        void SetFixedDistanceConstraint(size_t idx, size_t cameraIndex1, size_t cameraIndex2, float distance = 1.0f, float weight = 1.0f)
        {
            Check(mage_ba_set_fixed_distance_constraint(m_impl.get(), idx, cameraIndex1, cameraIndex2, distance, weight));
        }
        template <typename Q>
        void SetRelativeRotationConstraint(size_t idx, size_t cameraIndex1, size_t cameraIndex2, const Q& deltaRotation, float weight = 1.0f)
        {
            Check(mage_ba_set_relative_rotation_constraint(m_impl.get(), idx, cameraIndex1, cameraIndex2, detail::qdata(deltaRotation), weight));
        }
        template <typename V3, typename Q>
        void SetRelativeTransformConstraint(size_t idx, size_t cameraIndex1, size_t cameraIndex2, const V3& deltaPosition, const Q& deltaRotation, float weight)
        {
            Check(mage_ba_set_relative_transform_constraint(m_impl.get(), idx, cameraIndex1, cameraIndex2, detail::cdata(deltaPosition), detail::qdata(deltaRotation), weight));
        }

        void SetCurrentLambda(float userLambda) { Check(mage_ba_set_lambda(m_impl.get(), userLambda)); }
        float GetCurrentLambda() const { float v = 0; Check(mage_ba_get_lambda(m_impl.get(), &v)); return v; }

        // Runs an iteration of the solver for each provided Huber width.  Returns the average square error.
        // `huberWidthPerIteration` is anything with data()/size() (gsl::span<const float>, std::vector<float>).
        template <typename Span>
        float StepBundleAdjustment(const Span& huberWidthPerIteration, float maxErrorSquare, std::vector<unsigned int>& outliers)
        {
            return StepBundleAdjustment(huberWidthPerIteration.data(), static_cast<size_t>(huberWidthPerIteration.size()), maxErrorSquare, outliers);
        }
        float StepBundleAdjustment(const float* huberWidths, size_t count, float maxErrorSquare, std::vector<unsigned int>& outliers)
        {
            // The reference appends to `outliers` (BundlerLib.cpp:427-446) and the number of new entries is only known after the
            // call: the step keeps its complete list in the handle, and it is appended from there -- no capacity to guess.
            static_assert(sizeof(unsigned int) == sizeof(uint32_t), "outlier indices are 32-bit");
            size_t n = 0;
            float mse = 0;
            Check(mage_ba_step(m_impl.get(), huberWidths, count, maxErrorSquare, nullptr, 0, &n, &mse));
            if (n > 0) {
                const size_t old = outliers.size();
                outliers.resize(old + n);
                size_t got = 0;
                Check(mage_ba_get_outliers(m_impl.get(), reinterpret_cast<uint32_t*>(outliers.data() + old), n, &got));
            }
            return mse;
        }
        // kept for source compatibility with round-1 integrations; no capacity is needed any more
        void ReserveOutliers(size_t) {}

        template <typename V3, typename M3>
        void GetPose(size_t idx, V3&& position, M3&& orientation) const { Check(mage_ba_get_pose(m_impl.get(), idx, detail::mdata(position), detail::mdata(orientation))); }
        template <typename V3>
        void GetPoint(size_t idx, V3&& position) const { Check(mage_ba_get_point(m_impl.get(), idx, detail::mdata(position))); }

        mage_ba* Handle() const { return m_impl.get(); }   // for the bulk setters of mage_ba.h

    private:
        static void Check(mage_status s)
        {
            if (s != MAGE_OK) throw std::runtime_error(std::string("mage::BundlerLib: ") + mage_last_error());
        }
        struct Deleter { void operator()(mage_ba* h) const { mage_ba_destroy(h); } };
        std::unique_ptr<mage_ba, Deleter> m_impl;
        BundlerParameters m_bundlerParameters;
    };
}
