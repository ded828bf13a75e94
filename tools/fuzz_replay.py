#!/usr/bin/env python3
"""Replays ONE case of tools/fuzz_ba.py (same --seed, case number as printed by a mismatch; the same MAGE_BA_* environment) and prints the
LM trace of both sides per iteration -- chi2, lambda, their relative differences -- and the largest state difference.

    python tools/fuzz_replay.py <seed> <case>
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_ba as F
T = F.T
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
orig = T._compare_with_oracle
state = {"c": -1}
def patched(s, points_fixed, calls, rtol=1e-9):
    if state["c"] != target:
        return None, None            # skip the work, the rng stream does not depend on it
    from mageslam_amd.bundler import BundlerLib
    g, o = BundlerLib(points_fixed), T.OracleBundler(points_fixed)
    T._bulk(g, s); T.load_scene_bulk(o, s)
    og, oo = [], []
    for hubers, thr in calls:
        rg = g.StepBundleAdjustment(hubers, thr, og); ro = o.StepBundleAdjustment(hubers, thr, oo)
        print("call", hubers, thr, "mse", rg, ro, "outliers", len(og), len(oo), og == oo)
        for a, b in zip(g.trace(), o.trace()):
            print("   code %d/%d trials %d/%d chi_before %.15e chi_after %.15e | %.15e  rel %.2e   lam %.12e | %.12e rel %.2e" % (
                a["code"], b["code"], a["trials"], b["trials"], a.get("chi_before", float('nan')), a["chi_after"], b["chi_after"],
                abs(a["chi_after"] - b["chi_after"]) / max(b["chi_after"], 1e-300), a["lam"], b["lam"], abs(a["lam"] - b["lam"]) / b["lam"]))
    d = np.abs(g.poses_f64() - o.poses_f64()).max(); print("max pose diff", d, "max point diff", np.abs(g.points_f64() - o.points_f64()).max())
    return g, o
T._compare_with_oracle = patched
for c in range(target + 1):
    state["c"] = c
    F.one(seed * 100000 + c, rng)
