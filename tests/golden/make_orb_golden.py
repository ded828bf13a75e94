#!/usr/bin/env python3
"""Generates tests/golden/orb_*.npz -- run in the development container only.

Expected keypoints / blurred image / descriptors / matches come from the INDEPENDENT numpy implementation
(oracle/indep/orb_numpy.py), so the fixtures pin both the C oracle and the HIP path.  The reference cannot be run
(OpenCV is absent) and holds no ORB tests: "parity unpinned" with respect to the reference itself.
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mageslam_amd import frames  # noqa: E402
from oracle.indep import orb_numpy as N  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def base_pattern(patch):
    hdr = open(os.path.join(ROOT, "include", "mage_brief_patterns.h")).read()
    body = hdr.split(f"MAGE_BRIEF_BASE_{patch}[MAGE_BRIEF_PAIRS * 4] = {{")[1].split("}")[0]
    return np.array([int(v) for v in re.findall(r"-?\d+", body)], np.int64)


def main():
    b15 = base_pattern(15)
    # a 64x48 frame (few features: no suppression branch), a 160x120 frame and the full 640x480 pair (suppression + ANMS)
    cases = {"orb_64x48": frames.make_frame(101, 64, 48, n_rect=6, n_disc=8), "orb_160x120": frames.make_frame(102, 160, 120, n_rect=30, n_disc=40)}
    a, b = frames.frame_pair(7)
    cases["orb_640x480_a"] = a; cases["orb_640x480_b"] = b
    out = {}
    for name, img in cases.items():
        k, d, bl = N.detect(img, b15)
        h = __import__("hashlib").sha256(bl.tobytes()).hexdigest()
        out[name] = (img, k, d, h)
        print(name, img.shape, "keypoints", len(k), "blur sha", h[:12])
    m = N.match(out["orb_640x480_a"][2], out["orb_640x480_b"][2], 30, 1)
    print("matches", len(m))
    # RadiusMatch cases on the 640x480 pair: (radius, position override = A's keypoints mapped through the true homography)
    H = frames.small_homography(7)
    ka, kb = out["orb_640x480_a"][1], out["orb_640x480_b"][1]
    p3 = np.stack([ka[:, 0], ka[:, 1], np.ones(len(ka))], 0).astype(np.float64)
    q3 = H @ p3
    qpos = (q3[:2] / q3[2]).T.astype(np.float32)
    zo = np.zeros(len(ka), np.int64)
    rm_plain = N.radius_match(ka[:, :2], zo, out["orb_640x480_a"][2], kb[:, :2], np.zeros(len(kb), np.int64), out["orb_640x480_b"][2], 20.0, 30, 1)
    rm_over = N.radius_match(qpos, zo, out["orb_640x480_a"][2], kb[:, :2], np.zeros(len(kb), np.int64), out["orb_640x480_b"][2], 4.0, 40, 2)
    print("radius matches", len(rm_plain), len(rm_over))
    np.savez_compressed(os.path.join(HERE, "orb_frames.npz"), radius_qpos=qpos, radius_plain=rm_plain, radius_override=rm_over,
                        **{f"{n}_img": v[0] for n, v in out.items()}, **{f"{n}_kp": v[1] for n, v in out.items()},
                        **{f"{n}_desc": v[2] for n, v in out.items()}, **{f"{n}_blursha": np.array(v[3]) for n, v in out.items()},
                        matches_ab=m)


if __name__ == "__main__":
    main()
