// undistort_kernels.hip -- OrbFeatureDetector::UndistortKeypoints on the device (Image/OrbFeatureDetector.cpp:30-62).
//
// cv::undistortPoints of OpenCV 3.4.0 restated (modules/imgproc/src/undistort.cpp, cvUndistortPoints): per point, in float64,
//     x = (u - cx) / fx ... five iterations of  x <- (x0 - deltaX(x, y)) * icdist(r2)  ... re-projection with P, round to f32.
// One thread per keypoint, ~150 flops each: launch latency is the whole cost at 440 keypoints, so the batched form works on
// the device-resident output of the extractor and adds one short kernel to the frame pipeline instead of a host round trip.
// Floating-point contraction is OFF so that the float32 results equal the CPU restatement's bit for bit.
#include <hip/hip_runtime.h>
#include "orb_kernels.h"

#pragma clang fp contract(off)

namespace mage {
namespace {

__global__ __launch_bounds__(256) void k_undistort(mage_keypoint* __restrict__ kp, const int* __restrict__ counts, int capacity, int count_single,
                                                   UndistortConsts U)
{
    const int frame = blockIdx.y;
    const int n = counts ? min(counts[frame], capacity) : count_single;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    mage_keypoint* p = kp + (size_t)frame * capacity + i;
    double x = (double)p->x, y = (double)p->y;
    x = (x - U.cx) * U.ifx;
    y = (y - U.cy) * U.ify;
    const double x0 = x, y0 = y;
    const double* k = U.k;
#pragma unroll 1
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = U.RR[0] * x + U.RR[1] * y + U.RR[2];
    const double yy = U.RR[3] * x + U.RR[4] * y + U.RR[5];
    const double ww = 1. / (U.RR[6] * x + U.RR[7] * y + U.RR[8]);
    p->x = (float)(xx * ww);
    p->y = (float)(yy * ww);
}

}  // namespace

void undistort_launch(mage_keypoint* kp, const int* counts, int n_frames, int capacity, int count_single, const UndistortConsts& U, hipStream_t st)
{
    const int per_frame = counts ? capacity : count_single;
    if (per_frame <= 0 || n_frames <= 0) return;
    hipLaunchKernelGGL(k_undistort, dim3((per_frame + 255) / 256, n_frames), dim3(256), 0, st, kp, counts, capacity, count_single, U);
}

}  // namespace mage
