"""Probe of the UPSTREAM OpenCV routines the ORB / matching oracle restates: wherever a
`cv2` happens to be importable (it is not in the build image), the oracle's 8-bit Gaussian blur, the radius matcher's distance
and the undistortion are compared with it, and the OpenCV version is printed for DESIGN.md section 2.  Everywhere else the
tests SKIP -- nothing depends on cv2.  (cv2's stock ORB is not this fork: different selection / ANMS; not a parity target.)"""
import numpy as np
import pytest

cv2 = pytest.importorskip("cv2", reason="no OpenCV in this environment: the upstream probe cannot run (parity stays unpinned there)")

from mageslam_amd import frames
from oracle import oracle as O


def test_gaussian_blur_u8_matches_opencv():
    """cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) on CV_8U (OpenCVModified.cpp:853-865): the oracle pins the taps
    cvRound(256 * getGaussianKernel(7, 2)) and 16-bit fixed point; OpenCV >= 3.4 uses the same fixed-point path for u8."""
    img = frames.make_frame(5, 160, 120)
    _, _, blur = O.orb_detect(img, O.OrbParams.defaults(), want_blur=True)
    ref = cv2.GaussianBlur(img, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
    print("OpenCV", cv2.__version__, "max |oracle - cv2| =", int(np.abs(blur.astype(int) - ref.astype(int)).max()))
    assert np.array_equal(blur, ref)


def test_hamming_radius_match_distance_matches_opencv():
    """cv::BFMatcher(NORM_HAMMING).radiusMatch distances (FeatureMatcher.cpp:117-119) against the oracle's popcount."""
    rng = np.random.default_rng(3)
    A = rng.integers(0, 256, (40, 32), dtype=np.uint8); B = rng.integers(0, 256, (50, 32), dtype=np.uint8)
    B[:10] = A[:10]; B[3, 0] ^= 0x0F
    bf = cv2.BFMatcher(cv2.NORM_HAMMING)
    ref = bf.radiusMatch(A, B, 120.0)
    want = np.unpackbits(A[:, None, :] ^ B[None, :, :], axis=2).sum(axis=2)
    for qi, lst in enumerate(ref):
        for m in lst:
            assert int(m.distance) == int(want[qi, m.trainIdx])


def test_undistort_points_matches_opencv():
    """cv::undistortPoints with the 5-coefficient model (OrbFeatureDetector.cpp:30-62): five fixed-point iterations in float64."""
    K = np.array([[500, 0, 320], [0, 500, 240], [0, 0, 1]], np.float32)
    dist = np.array([0.1, -0.05, 0.001, -0.002, 0.01], np.float32)
    pts = np.random.default_rng(2).uniform([10, 10], [630, 470], (64, 2)).astype(np.float32)
    kps = np.zeros(len(pts), O.KEYPOINT_DTYPE); kps["x"] = pts[:, 0]; kps["y"] = pts[:, 1]
    got = O.undistort_keypoints(kps, O.UndistortParams.make(K, dist, K))
    ref = cv2.undistortPoints(pts.reshape(-1, 1, 2), K, dist, P=K).reshape(-1, 2)
    print("OpenCV", cv2.__version__, "max |oracle - cv2| =", float(np.abs(np.stack([got["x"], got["y"]], 1) - ref).max()))
    np.testing.assert_allclose(np.stack([got["x"], got["y"]], 1), ref, atol=2e-3)       # 3.4.0 iterates 5 times, later versions until convergence
