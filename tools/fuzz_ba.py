#!/usr/bin/env python3
"""Differential campaign for the bundle adjustment beyond the suite's fixed cases: random graph shapes (a handful to ~60 cameras, so
the one-launch pose solver, the small-problem path and the large-problem path all get traffic; well-posed graphs only -- with points
seen once the damping alone fixes their depth and HIP and oracle drift apart by conditioning, in every launch sequence alike),
shuffled / thinned / duplicated observations, random fixed flags, tethers, points fixed or free, several calls with shrinking outlier thresholds -- HIP against the
CPU oracle with the comparisons of tests/test_ba_gpu.py (outlier lists identical, LM trace identical, state to 1e-8).

    python tools/fuzz_ba.py [--cases 200] [--seed 1] [--big]
"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
from mageslam_amd import scene  # noqa: E402
import test_ba_gpu as T  # noqa: E402


def one(case, rng, big=False):
    n_cams = int(rng.choice([2, 3, 5, 8, 13, 20, 30, 45, 60]))
    n_pts = int(rng.integers(6, 40 * max(n_cams, 2)))
    K = int(rng.integers(2, min(n_cams, 8) + 1))
    if big:          # maps of more than 2 048 Schur blocks (the large-problem launch sequence with k_schur_stream), tiled dense solve
        n_cams = int(rng.choice([110, 150, 190, 230]))
        K = int(rng.choice([8, 12, 16, 20]))
        n_pts = int(rng.integers(6 * n_cams, 14 * n_cams))          # every point seen at least twice: single views leave the depth to the damping alone
    s = scene.make_scene(n_cams=n_cams, n_pts=n_pts, n_obs=n_pts * K, seed=0x5EED9000 + case, fixed=(), outlier_frac=float(rng.choice([0.0, 0.02, 0.1])))
    idx = rng.permutation(s.n_obs)
    idx = idx[rng.random(s.n_obs) > rng.choice([0.0, 0.05, 0.1])]
    if len(idx) == 0:
        return "skipped (no observation left)"
    if rng.random() < 0.5:
        dup = rng.choice(idx, size=max(1, len(idx) // 8))
        idx = np.concatenate([idx, dup])
    s.obs_uv, s.obs_cam, s.obs_pt, s.obs_info = s.obs_uv[idx].copy(), s.obs_cam[idx], s.obs_pt[idx], s.obs_info[idx]
    s.n_obs = len(idx)
    fixed = rng.random(n_cams) < rng.choice([0.0, 0.2, 0.5])
    if n_cams > 1:
        fixed[int(rng.integers(0, n_cams))] = True
    s.cam_fixed = fixed
    tethered = n_cams >= 3 and rng.random() < 0.25
    if tethered:
        s.tethers = scene.make_tethers(s, n_dist=int(rng.integers(0, 3)), n_rot=int(rng.integers(0, 3)), n_xf=int(rng.integers(1, 3)), seed=0x7E7E9000 + case)
    points_fixed = rng.random() < 0.25
    hub = [float(rng.choice([0.9, 1.8, 4.0]))] * int(rng.integers(1, 5))
    # a free camera seen through fewer than three points has a pose the observations do not determine (damping and tethers alone hold
    # it): rounding differences are then amplified without bound, and neither side is "right" -- not a parity case
    per_cam = np.bincount(s.obs_cam.astype(np.int64), minlength=n_cams)
    if (~fixed).any() and per_cam[~fixed].min() < 3:
        return "skipped (a free camera with fewer than three observations)"
    calls = [(hub, float(rng.choice([1e30, 30.0, 9.0])))] + [([0.9], float(rng.choice([1e30, 16.0, 5.0])))] * int(rng.integers(0, 3))
    try:
        T._compare_with_oracle(s, points_fixed, calls, rtol=1e-6 if tethered else 1e-8)
    except AssertionError:
        print("   shape:", dict(n_cams=n_cams, n_pts=n_pts, K=K, n_obs=s.n_obs, fixed=int(fixed.sum()), tethered=tethered, points_fixed=points_fixed, calls=calls), flush=True)
        # which switch, if any, makes the same scene agree: the structure build's twin, the classic outlier pass, a looser tolerance
        for name, env, tol in (("MAGE_BA_BUILD=host", {"MAGE_BA_BUILD": "host"}, None), ("MAGE_BA_BUILD=device", {"MAGE_BA_BUILD": "device"}, None),
                               ("MAGE_BA_CONSERVATIVE=1", {"MAGE_BA_CONSERVATIVE": "1"}, None), ("rtol 1e-5", {}, 1e-5)):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                T._compare_with_oracle(s, points_fixed, calls, rtol=tol or (1e-6 if tethered else 1e-8))
                verdict = "agrees"
            except AssertionError as e:
                verdict = "still differs: " + str(e)[:120].replace("\n", " ")
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            print(f"   with {name}: {verdict}", flush=True)
        raise
    return None


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--cases", type=int, default=200); ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", action="store_true", help="110-230 cameras, 8-20 views per point: more than 2 048 Schur blocks")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    bad = skipped = 0
    t0 = time.time()
    for c in range(a.cases):
        try:
            r = one(a.seed * 100000 + c, rng, a.big)
            skipped += r is not None
        except AssertionError:
            bad += 1
            print("BA MISMATCH case", c, traceback.format_exc().strip().splitlines()[-3:], flush=True)
    print(f"{a.cases} BA cases ({skipped} skipped), {bad} mismatches, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
