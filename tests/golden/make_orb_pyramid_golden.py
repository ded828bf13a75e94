#!/usr/bin/env python3
"""Generates tests/golden/orb_pyramid.npz -- NumLevels > 1 fixtures (development container only).

Frames are the committed ones of orb_frames.npz; expected keypoints (level coordinates, response, octave), float32 image
coordinates, descriptors and (with orientation) angles come from the INDEPENDENT numpy implementation (oracle/indep/orb_numpy.py):
cv::resize INTER_LINEAR restated in 11-bit fixed point, per-level quotas, every level blurred as an isolated image (the pinned
deviation, DESIGN.md section 9 / oracle/orb_oracle.c).
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.indep import orb_numpy as N  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = {"l3_160x120": ("orb_160x120", dict(nlevels=3)),
         "l2_s12_640x480": ("orb_640x480_a", dict(nlevels=2, scale_factor=1.2)),
         "l2_oriented_160x120": ("orb_160x120", dict(nlevels=2, use_orientation=1)),
         "l4_p31_160x120": ("orb_160x120", dict(nlevels=4, patch_size=31, nfeatures=300))}


def base_pattern(patch):
    hdr = open(os.path.join(ROOT, "include", "mage_brief_patterns.h")).read()
    body = hdr.split(f"MAGE_BRIEF_BASE_{patch}[MAGE_BRIEF_PAIRS * 4] = {{")[1].split("}")[0]
    return np.array([int(v) for v in re.findall(r"-?\d+", body)], np.int64)


def main():
    g = np.load(os.path.join(HERE, "orb_frames.npz"))
    out = {"resize_src": g["orb_160x120_img"], "resize_107x80": N.resize_linear(g["orb_160x120_img"], 107, 80),
           "resize_53x91": N.resize_linear(g["orb_160x120_img"], 53, 91)}
    for key, (name, kw) in CASES.items():
        r = N.detect(g[name + "_img"], base_pattern(kw.get("patch_size", 15)), **kw)
        P = dict(N.DEFAULTS); P.update(kw)
        out[key + "_kp"], out[key + "_desc"], out[key + "_xy"] = r[0], r[1], N.scaled_xy(r[0], P)
        if len(r) == 4:
            out[key + "_angle"] = r[3]
        print(key, len(r[0]), "per level", np.bincount(r[0][:, 3], minlength=kw["nlevels"]))
    # ORB-9: patch sizes without a pre-rotated table take the cv::RNG random pattern (angle 0)
    for patch in (21, 9):
        k, d, _ = N.detect(g["orb_160x120_img"], None, patch_size=patch)
        out[f"rand{patch}_kp"], out[f"rand{patch}_desc"], out[f"rand{patch}_pattern"] = k, d, N.random_pattern(patch)
        print("random pattern patch", patch, len(k))
    np.savez_compressed(os.path.join(HERE, "orb_pyramid.npz"), **out)


if __name__ == "__main__":
    main()
