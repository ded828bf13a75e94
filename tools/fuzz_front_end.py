#!/usr/bin/env python3
"""Differential campaign beyond the test suite's fixed cases: random frames / settings through the ORB detector and random descriptor
sets through the matchers, HIP against the CPU oracle, bit for bit.  Prints one line per failure and a summary.

    python tools/fuzz_front_end.py [--cases 400] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401  (torch's HIP runtime first, see mageslam_amd/_lib.py)
from mageslam_amd import frames  # noqa: E402
from mageslam_amd.orb import Matcher, OrbDetector  # noqa: E402
from oracle import oracle as O  # noqa: E402

OKW = {"feature_factor_anms": "feature_factor", "feature_strength_anms": "feature_strength", "strong_response_anms": "strong_response",
       "min_robust_factor": "min_robust", "max_robust_factor": "max_robust", "num_cells_x": "cells_x", "num_cells_y": "cells_y"}


def orb_case(rng, case):
    w, h = int(rng.integers(24, 700)), int(rng.integers(24, 520))
    kind = case % 6
    if kind == 4:
        img = rng.integers(0, 256, (h, w)).astype(np.uint8)
    elif kind == 5:
        img = (rng.integers(0, 2, (h, w)) * int(rng.integers(30, 255))).astype(np.uint8)
    else:
        img = frames.make_frame(5000 + case, w, h, n_rect=int(rng.integers(3, 200)), n_disc=int(rng.integers(3, 300)), noise=float(rng.choice([0.0, 1.0, 2.0, 4.0])))
    patch = int(rng.choice([15, 31, 9, 21])) if case % 4 == 0 else int(rng.choice([15, 31]))
    kw = dict(nfeatures=int(rng.integers(4, 3000 if case % 7 == 0 else 700)), fast_threshold=int(rng.integers(1, 256) if case % 9 == 0 else rng.integers(1, 60)), num_cells_x=int(rng.integers(1, 48)),
              num_cells_y=int(rng.integers(1, 48)), gaussian_kernel_size=int(rng.choice([1, 3, 5, 7, 7, 7, 9])), patch_size=patch,
              nlevels=int(rng.integers(1, 5)) if case % 3 == 0 else 1, scale_factor=float(rng.choice([1.2, 1.5, 2.0])), use_orientation=int(case % 5 == 0),
              feature_factor_anms=float(rng.choice([1.0, 1.5, 2.5])), feature_strength_anms=float(rng.choice([0.5, 0.9, 1.0, 1.2])),
              strong_response_anms=int(rng.integers(5, 60)), min_robust_factor=float(rng.choice([1.0, 1.1])), max_robust_factor=float(rng.choice([2.0, 2.2, 3.0])))
    k, d = OrbDetector(**kw).DetectAndCompute(img)
    ko, do = O.orb_detect(img, O.OrbParams.defaults(**{OKW.get(a, a): b for a, b in kw.items()}))
    ok = len(k) == len(ko) and all(np.array_equal(k[f], ko[f]) for f in ("x", "y", "response", "octave", "size", "angle", "class_id")) and np.array_equal(d, do)
    return ok, (w, h, kw), len(ko)


def match_case(rng, case, mt):
    nA, nB = int(rng.integers(0, 700)), int(rng.integers(0, 700))
    centres = rng.integers(0, 256, (max(1, int(rng.integers(1, 20))), 32)).astype(np.uint8)

    def draw(n):
        d = centres[rng.integers(0, len(centres), n)].copy()
        d ^= np.packbits((rng.random((n, 32, 8)) < rng.choice([0.0, 0.01, 0.05])).astype(np.uint8), axis=2).reshape(n, 32)
        fresh = rng.random(n) < 0.3
        d[fresh] = rng.integers(0, 256, (int(fresh.sum()), 32)).astype(np.uint8)
        return d

    A, B = draw(nA), draw(nB)
    md, mdiff = int(rng.choice([0, 10, 30, 64, 256])), int(rng.choice([0, 1, 2, 5]))
    m = mt.Match(A, B, None, None, md, mdiff)
    mo = O.match(A, B, md, mdiff)
    ok = len(m) == len(mo) and (len(mo) == 0 or (np.array_equal(m["queryIdx"], mo["queryIdx"]) and np.array_equal(m["trainIdx"], mo["trainIdx"]) and np.array_equal(m["distance"], mo["distance"])))
    return ok, (nA, nB, md, mdiff), len(mo)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=400); ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    mt = Matcher()
    bad = 0
    t0 = time.time()
    nk, full, nm = [], 0, []
    for c in range(a.cases):
        ok, what, n = orb_case(rng, c)
        nk.append(n); full += int(n == what[2]["nfeatures"] or n >= what[2]["nfeatures"] * 0.9)
        if not ok:
            bad += 1; print("ORB MISMATCH case", c, what, flush=True)
        ok, what, n = match_case(rng, c, mt)
        nm.append(n)
        if not ok:
            bad += 1; print("MATCH MISMATCH case", c, what, flush=True)
    print(f"{a.cases} ORB cases (keypoints per case: mean {np.mean(nk):.0f}, {int(np.sum(np.array(nk) > 0))} non-empty, {full} at the feature budget = suppression ran) + "
          f"{a.cases} matcher cases (matches per case: mean {np.mean(nm):.0f}), {bad} mismatches, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
