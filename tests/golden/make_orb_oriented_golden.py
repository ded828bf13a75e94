#!/usr/bin/env python3
"""Generates tests/golden/orb_oriented.npz -- UseOrientation fixtures (development container only).

The frames are the committed ones of orb_frames.npz; expected keypoints, ICAngles orientations and rotated-BRIEF descriptors
come from the INDEPENDENT numpy implementation (oracle/indep/orb_numpy.py) with use_orientation=True: border = ceil(7 sqrt 2),
intensity-centroid angle through cv::fastAtan2's polynomial, descriptor row cvRound(angle / 12) % 30 of the pre-rotated table.
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.indep import orb_numpy as N  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def base_pattern(patch):
    hdr = open(os.path.join(ROOT, "include", "mage_brief_patterns.h")).read()
    body = hdr.split(f"MAGE_BRIEF_BASE_{patch}[MAGE_BRIEF_PAIRS * 4] = {{")[1].split("}")[0]
    return np.array([int(v) for v in re.findall(r"-?\d+", body)], np.int64)


def main():
    g = np.load(os.path.join(HERE, "orb_frames.npz"))
    out = {}
    for name, patch in (("orb_160x120", 15), ("orb_640x480_a", 15), ("orb_160x120", 31)):
        k, d, _, ang = N.detect(g[name + "_img"], base_pattern(patch), use_orientation=True, patch_size=patch)
        key = f"{name}_p{patch}"
        out[key + "_kp"], out[key + "_desc"], out[key + "_angle"] = k, d, ang
        print(key, len(k), "rows used", len(np.unique(np.rint(ang / np.float32(12)).astype(int) % 30)))
    np.savez_compressed(os.path.join(HERE, "orb_oriented.npz"), **out)


if __name__ == "__main__":
    main()
