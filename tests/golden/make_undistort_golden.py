#!/usr/bin/env python3
"""Generates tests/golden/orb_undistort.npz -- UndistortKeypoints fixtures (development container only).

Points are the committed keypoints of the 640x480 frame A plus the image corners and the principal point; expected values come
from the INDEPENDENT numpy restatement of cv::undistortPoints (oracle/indep/orb_numpy.py), for a Poly3k (5 coefficients) and a
Rational6k (8 coefficients) calibration.  OpenCV itself is absent: parity with the reference's OpenCV 3.4.0 stays unpinned.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.indep import orb_numpy as N  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
K = np.array([[498.6, 0, 321.4], [0, 501.2, 238.7], [0, 0, 1]], np.float32)
P = np.array([[470.0, 0, 320.0], [0, 470.0, 240.0], [0, 0, 1]], np.float32)
CASES = {"poly3k": np.array([-0.281, 0.0912, 8.1e-4, -4.3e-4, -0.0137], np.float32),
         "rational6k": np.array([0.523, -0.214, 9.0e-4, 2.2e-4, 0.0481, 0.812, -0.0973, 0.0214], np.float32)}


def main():
    g = np.load(os.path.join(HERE, "orb_frames.npz"))
    xy = g["orb_640x480_a_kp"][:, :2].astype(np.float32)
    xy = np.concatenate([xy, np.array([[0, 0], [639, 0], [0, 479], [639, 479], [321.4, 238.7]], np.float32)])
    out = {"xy": xy, "K": K, "P": P}
    for name, d in CASES.items():
        out["dist_" + name] = d
        out["exp_" + name] = N.undistort_points(xy, K, d, P)
        print(name, "max shift", np.abs(out["exp_" + name] - xy).max())
    np.savez_compressed(os.path.join(HERE, "orb_undistort.npz"), **out)


if __name__ == "__main__":
    main()
