/*
 * mage_orb.h -- C ABI of the MI355X ORB front-end (libmageslam_hip.so).
 *
 * Drop-in boundary for the reference's detector:
 *     Core/MAGESLAM/Source/Image/OpenCVModified.h:64-173   class OrbDetector (ctor :68-82, DetectAndCompute :85-88)
 *     Core/MAGESLAM/Source/Image/OpenCVModified.cpp:771-886 DetectAndCompute  (FAST-9/16 + 3x3 NMS, border cull,
 *                                                           RetainBestFeatures, ANMS, GaussianBlur, BRIEF-256)
 *     Core/MAGESLAM/Source/Image/OrbFeatureDetector.cpp:84-100 Process (the caller; undistortion stays host-side)
 * Output records are bit-compatible with cv::KeyPoint (28 bytes) and mage::ORBDescriptor (32 bytes,
 * Image/ORBDescriptor.h:12-23), written into caller-owned fixed-capacity buffers exactly as
 * ImageData::Insert does (Image/ImageData.h:65-70: truncation to the capacity).
 *
 * Where the reference's result is implementation-defined (std::nth_element order) the order is pinned:
 * keypoints that survive without suppression stay in raster order; after ANMS they are ordered by
 * (suppression radius desc, response desc, raster index asc).  See DESIGN.md section 6.
 */
#ifndef MAGE_ORB_H
#define MAGE_ORB_H

#include <stddef.h>
#include <stdint.h>

#include "mage_ba.h"   /* mage_status, mage_last_error */

#ifdef __cplusplus
extern "C" {
#endif

/* cv::KeyPoint */
typedef struct mage_keypoint {
    float x, y;        /* pt */
    float size;        /* patchSize * layerScale */
    float angle;       /* 0 when UseOrientation is off */
    float response;    /* FAST score */
    int   octave;
    int   class_id;    /* -1 */
} mage_keypoint;

/* OrbDetector ctor arguments (OpenCVModified.h:68-82) = FeatureExtractorSettings (MageSettings.h:151-167) + device. */
typedef struct mage_orb_params {
    unsigned gaussian_kernel_size;   /* 7 */
    unsigned nfeatures;              /* 440 */
    float    scale_factor;           /* 1.5 (unused with one level) */
    unsigned nlevels;                /* 1 (default) .. 16: cv::resize(INTER_LINEAR) pyramid with per-level quotas (OpenCVModified.cpp:659-669, 793-841);
                                        every level is blurred as an ISOLATED image -- the reference's in-place ROI blur reads the neighbouring level or
                                        uninitialised memory at the borders, which cannot be restated (pinned deviation) */
    unsigned patch_size;             /* 15 or 31: pre-rotated tables; 2 .. 127 otherwise: MakeRandomPattern (cv::RNG, OpenCVModified.cpp:551-560),
                                        then only with use_orientation = 0 (its rotation would go through libm: MAGE_ERR_UNSUPPORTED) */
    unsigned fast_threshold;         /* 4 */
    int      use_orientation;        /* 0 (default) or 1: ICAngles orientation + rotated BRIEF rows (OpenCVModified.cpp:399-437, 523-532) */
    float    feature_factor_anms;    /* 1.5 */
    float    feature_strength_anms;  /* 0.9 */
    int      strong_response_anms;   /* 20 */
    float    min_robust_factor;      /* 1.1 */
    float    max_robust_factor;      /* 2.0 */
    int      num_cells_x, num_cells_y; /* 32, 32 */
    int      device;                 /* HIP ordinal, -1 = current */
} mage_orb_params;

typedef struct mage_orb mage_orb;

mage_status mage_orb_default_params(mage_orb_params* p);
mage_status mage_orb_create(const mage_orb_params* params, mage_orb** out);
void        mage_orb_destroy(mage_orb* h);

/* OrbDetector::DetectAndCompute on one CV_8UC1 image held in HOST memory (row pitch `stride` bytes).
 * Writes up to `capacity` keypoints / 32-byte descriptors; *count receives the number written. */
mage_status mage_orb_detect(mage_orb* h, const uint8_t* image, int width, int height, int stride,
                            mage_keypoint* keypoints, uint8_t* descriptors32, int capacity, int* count);

/* Batched form: n_frames images of identical size.  `images` may be a host pointer (images_on_device = 0) or a
 * device pointer in the handle's HBM (images_on_device = 1), frames packed with pitch `stride` and
 * `frame_stride` bytes between frames.  Outputs are host buffers of n_frames x capacity records;
 * counts[n_frames].  Every frame is processed independently (stereo pairs, frame queues). */
mage_status mage_orb_detect_batch(mage_orb* h, const uint8_t* images, int images_on_device, int n_frames, int width, int height,
                                  int stride, size_t frame_stride, mage_keypoint* keypoints, uint8_t* descriptors32,
                                  int capacity, int* counts);

/* Same, but the results stay in HBM: pointers are device pointers owned by the handle, valid until the next
 * call on the handle (keypoints: n_frames x capacity mage_keypoint; descriptors: n_frames x capacity x 32;
 * counts: n_frames int).  Used to chain into mage_match_bf_batch_device without a host round trip. */
mage_status mage_orb_detect_batch_device(mage_orb* h, const uint8_t* images_device, int n_frames, int width, int height,
                                         int stride, size_t frame_stride, int capacity, const mage_keypoint** keypoints_device,
                                         const uint8_t** descriptors_device, const int** counts_device);

/* OrbFeatureDetector::UndistortKeypoints (Image/OrbFeatureDetector.cpp:30-62): cv::undistortPoints(points, cameraMatrix,
 * distCoeffs, noArray(), newCameraMatrix) of OpenCV 3.4.0 on the keypoint coordinates -- normalise with the distorted
 * camera matrix, five fixed-point iterations of the inverse radial / tangential model in float64, re-project with the
 * undistorted camera matrix, round to float32.  Matrices are row-major 3x3 (cv::Matx33f memory order); coefficients in
 * OpenCV order k1 k2 p1 p2 k3 [k4 k5 k6], n_dist in {4, 5, 8} (the reference passes 5 for Poly3k, 8 for Rational6k:
 * Device/CameraCalibration.cpp:110-125).  Only x and y of each keypoint change.  Thin-prism / tilt terms (12, 14
 * coefficients) are not in the reference's calibration models and are refused. */
typedef struct mage_undistort_params {
    float camera_matrix[9];
    float dist_coeffs[8];
    int   n_dist;
    float new_camera_matrix[9];
} mage_undistort_params;
/* keypoints: HOST array of `count` records, rewritten in place. */
mage_status mage_orb_undistort_keypoints(mage_orb* h, mage_keypoint* keypoints, int count, const mage_undistort_params* params);
/* keypoints_device / counts_device: the buffers mage_orb_detect_batch_device returned (n_frames x capacity records, n_frames counts);
 * rewritten in place in HBM, asynchronously on the handle's stream (the next call on the handle orders after it). */
mage_status mage_orb_undistort_keypoints_device(mage_orb* h, const mage_keypoint* keypoints_device, const int* counts_device, int n_frames,
                                                int capacity, const mage_undistort_params* params);

/* Stage outputs of the most recent single-frame / first frame of a batch, for parity tests:
 * FAST score map (width x height u8) and blurred image (width x height u8). */
mage_status mage_orb_debug_read(mage_orb* h, uint8_t* score_map, uint8_t* blurred);

/* Per-stage HIP-event timings of the most recent batch call (milliseconds).  Off by default (all zero): the event packets between
 * the kernels cost more than a one-frame batch's kernels are apart.  With the 7-tap Gaussian the blur runs inside the FAST launch:
 * fast_ms then includes it and blur_ms is ~0. */
typedef struct mage_orb_profile {
    double fast_ms, select_ms, blur_ms, brief_ms, total_ms;
    int    n_frames;
} mage_orb_profile;
mage_status mage_orb_enable_profile(mage_orb* h, int on);
mage_status mage_orb_get_profile(const mage_orb* h, mage_orb_profile* out);

#ifdef __cplusplus
}
#endif
#endif /* MAGE_ORB_H */
