// OrbDetector.h -- C++ shim with the reference's class names and argument order over the MI355X C ABI (mage_orb.h).
//
// Replaces the two classes a MAGE-SLAM build instantiates for its front end:
//   ::OrbDetector                 Core/MAGESLAM/Source/Image/OpenCVModified.h:64-88   (constructor: the 14 settings in the same
//                                 order; DetectAndCompute)
//   mage::OrbFeatureDetector      Core/MAGESLAM/Source/Image/OrbFeatureDetector.h:32-47 (Process = DetectAndCompute + UndistortKeypoints,
//                                 OrbFeatureDetector.cpp:64-83, :30-62)
// The reference passes cv::Mat / cv::KeyPoint / ORBDescriptor / ImageData; those are plain records or views over them, so the shim
// is a set of templates over "anything shaped like it": an image needs .data / .cols / .rows / .step (cv::Mat has exactly these),
// keypoints are any 28-byte record with cv::KeyPoint's layout (pt.x, pt.y, size, angle, response, octave, class_id = mage_keypoint),
// descriptors any 32-byte record (ORBDescriptor's storage).  With OpenCV on the include path the original call sites bind to these
// templates as written; without it the raw-pointer overloads work (tools/shim_orb_match.cpp compiles them with the host compiler alone).
//
// Error behaviour: the reference asserts; the shim throws std::runtime_error carrying mage_last_error().
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "mage_orb.h"

namespace mage
{
    namespace shim
    {
        inline void Check(mage_status s, const char* who)
        {
            if (s != MAGE_OK) throw std::runtime_error(std::string(who) + ": " + mage_last_error());
        }
    }
}

class OrbDetector
{
public:
    OrbDetector(unsigned int gaussianKernelSize, unsigned int nfeatures, float scaleFactor, unsigned int nlevels, unsigned int patchSize,
                unsigned int fastThreshold, bool useOrientation, float featureFactorANMS, float featureStrengthANMS, int strongResponseANMS,
                float minRobustFactor, float maxRobustFactor, int numCellsX, int numCellsY, int device = -1)
        : m_nfeatures(nfeatures)
    {
        mage_orb_params p{ gaussianKernelSize, nfeatures, scaleFactor, nlevels, patchSize, fastThreshold, useOrientation ? 1 : 0, featureFactorANMS,
                           featureStrengthANMS, strongResponseANMS, minRobustFactor, maxRobustFactor, numCellsX, numCellsY, device };
        mage_orb* h = nullptr;
        mage::shim::Check(mage_orb_create(&p, &h), "OrbDetector");
        m_impl.reset(h);
    }
    OrbDetector(const OrbDetector&) = delete;
    OrbDetector& operator=(const OrbDetector&) = delete;

    // Compute the ORB features and descriptors on an image (CV_8UC1).  The reference fills an ImageData sized for nfeatures; here the
    // two vectors are resized to the number of features found (<= nfeatures).
    template <typename Mat, typename KeyPointVec, typename DescriptorVec>
    void DetectAndCompute(const Mat& image, KeyPointVec& keypoints, DescriptorVec& descriptors)
    {
        DetectAndCompute(static_cast<const uint8_t*>(image.data), static_cast<int>(image.cols), static_cast<int>(image.rows),
                         static_cast<int>(static_cast<size_t>(image.step)), keypoints, descriptors);
    }
    template <typename KeyPointVec, typename DescriptorVec>
    void DetectAndCompute(const uint8_t* image, int width, int height, int strideBytes, KeyPointVec& keypoints, DescriptorVec& descriptors)
    {
        static_assert(sizeof(typename KeyPointVec::value_type) == sizeof(mage_keypoint), "keypoints must have cv::KeyPoint's 28-byte layout");
        static_assert(sizeof(typename DescriptorVec::value_type) == 32, "descriptors are 32 bytes");
        keypoints.resize(m_nfeatures);
        descriptors.resize(m_nfeatures);
        int count = 0;
        mage::shim::Check(mage_orb_detect(m_impl.get(), image, width, height, strideBytes, reinterpret_cast<mage_keypoint*>(keypoints.data()),
                                          reinterpret_cast<uint8_t*>(descriptors.data()), static_cast<int>(m_nfeatures), &count), "OrbDetector::DetectAndCompute");
        keypoints.resize(static_cast<size_t>(count));
        descriptors.resize(static_cast<size_t>(count));
    }

    mage_orb* Handle() const { return m_impl.get(); }   // for the batched / device-resident entry points of mage_orb.h

private:
    struct Deleter { void operator()(mage_orb* h) const { mage_orb_destroy(h); } };
    std::unique_ptr<mage_orb, Deleter> m_impl;
    unsigned int m_nfeatures;
};

namespace mage
{
    // FeatureExtractorSettings (Core/MAGESLAM/Source/MageSettings.h:151-167), same member names and defaults
    struct FeatureExtractorSettings
    {
        unsigned int GaussianKernelSize = 7;
        unsigned int NumFeatures = 440;
        float ScaleFactor = 1.5f;
        unsigned int NumLevels = 1;
        unsigned int PatchSize = 15;
        unsigned int FastThreshold = 4;
        bool UseOrientation = false;
        float FeatureFactor = 1.5f;
        float FeatureStrength = 0.9f;
        int StrongResponse = 20;
        float MinRobustnessFactor = 1.1f;
        float MaxRobustnessFactor = 2.0f;
        int NumCellsX = 32;
        int NumCellsY = 32;
    };

    class OrbFeatureDetector
    {
    public:
        explicit OrbFeatureDetector(const FeatureExtractorSettings& s, int device = -1)
            : m_detector(s.GaussianKernelSize, s.NumFeatures, s.ScaleFactor, s.NumLevels, s.PatchSize, s.FastThreshold, s.UseOrientation, s.FeatureFactor,
                         s.FeatureStrength, s.StrongResponse, s.MinRobustnessFactor, s.MaxRobustnessFactor, s.NumCellsX, s.NumCellsY, device)
        {
        }

        // Process (OrbFeatureDetector.cpp:64-83): detect + describe, then move the keypoints to where the undistorted camera sees them.
        // cameraMatrix / newCameraMatrix: 9 floats row-major (cv::Matx33f memory order); distCoeffs: k1 k2 p1 p2 k3 [k4 k5 k6]
        // (CameraCalibration::GetCVDistortionCoeffs: 5 for Poly3k, 8 for Rational6k).
        template <typename KeyPointVec, typename DescriptorVec>
        void Process(const float* cameraMatrix, const float* distCoeffs, int numDistCoeffs, const float* newCameraMatrix, const uint8_t* image, int width,
                     int height, int strideBytes, KeyPointVec& keypoints, DescriptorVec& descriptors)
        {
            m_detector.DetectAndCompute(image, width, height, strideBytes, keypoints, descriptors);
            UndistortKeypoints(keypoints, cameraMatrix, distCoeffs, numDistCoeffs, newCameraMatrix);
        }

        template <typename KeyPointVec>
        void UndistortKeypoints(KeyPointVec& inoutKeypoints, const float* cameraMatrix, const float* distCoeffs, int numDistCoeffs, const float* newCameraMatrix)
        {
            static_assert(sizeof(typename KeyPointVec::value_type) == sizeof(mage_keypoint), "keypoints must have cv::KeyPoint's 28-byte layout");
            mage_undistort_params p{};
            for (int i = 0; i < 9; ++i) { p.camera_matrix[i] = cameraMatrix[i]; p.new_camera_matrix[i] = newCameraMatrix[i]; }
            for (int i = 0; i < numDistCoeffs && i < 8; ++i) p.dist_coeffs[i] = distCoeffs[i];
            p.n_dist = numDistCoeffs;
            shim::Check(mage_orb_undistort_keypoints(m_detector.Handle(), reinterpret_cast<mage_keypoint*>(inoutKeypoints.data()),
                                                     static_cast<int>(inoutKeypoints.size()), &p), "OrbFeatureDetector::UndistortKeypoints");
        }

        ::OrbDetector& Detector() { return m_detector; }

    private:
        ::OrbDetector m_detector;
    };
}
