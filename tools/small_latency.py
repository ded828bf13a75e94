"""Latency of the small problems the reference solves every keyframe / every frame:
  * local bundle adjustment, BASELINE.json configs[2] (20 keyframes / 5k points / 50k observations): ms per LM iteration;
  * pose-only refinement (one free camera, ~200 fixed points, 4 iterations as TrackLocalMap::OptimizeCameraPose runs it):
    ms per call including create / set / destroy.
Run with MAGE_BA_NO_SMALL_PATH=1 for the large-problem path on the same inputs.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mageslam_amd import scene
from mageslam_amd.bundler import BundlerLib, load_scene


def main():
    out = {"small_path": "MAGE_BA_NO_SMALL_PATH" not in os.environ}
    s = scene.make_config("local", outlier_frac=0.02)
    def local_ba():
        # the reference's local BA (BundleAdjust.cpp:281-354): a bundler per run, 10 LM iterations, shrinking outlier threshold
        b = BundlerLib(False, device=0)
        load_scene(b, s, bulk=True)
        thr, o, r = 7.25, [], 0.0
        t0 = time.perf_counter()
        for _ in range(10):
            r = b.StepBundleAdjustment([0.9], thr, o); thr *= 0.95 * 0.95
        t1 = time.perf_counter()
        b.close()
        return t1 - t0, r, len(o)
    for _ in range(3):
        local_ba()
    n = 50
    ts = [local_ba() for _ in range(n)]
    out["local_ms_per_lm_iteration"] = 1e3 * sum(t for t, _, _ in ts) / (10 * n)
    out["local_rmse_px"] = float(np.sqrt(ts[-1][1])); out["local_outliers"] = ts[-1][2]
    # steady state: the LM iteration alone (structure built, no outlier removed, no re-seed of lambda)
    b = BundlerLib(False, device=0)
    load_scene(b, scene.make_config("local"), bulk=True)
    b.StepBundleAdjustment([0.9], 1e30, [])
    t0 = time.perf_counter()
    for _ in range(6):
        b.StepBundleAdjustment([0.9], 1e30, [])
    out["local_steady_ms_per_lm_iteration"] = 1e3 * (time.perf_counter() - t0) / 6
    out["local_steady_trials"] = [t["trials"] for t in b.trace()]
    b.close()
    n = 200
    # pose-only: one free camera observing 200 fixed points
    p = scene.make_scene(n_cams=1, n_pts=200, n_obs=200, seed=0x5EED0A77, fixed=())
    def call():
        g = BundlerLib(True, device=0)
        load_scene(g, p, bulk=True)
        r = g.StepBundleAdjustment([1.8] * 4, 1e30, [])
        g.close()
        return r
    for _ in range(5):
        call()
    t0 = time.perf_counter()
    for _ in range(n):
        r = call()
    out["pose_only_ms_per_call_4_iterations_incl_create_destroy"] = 1e3 * (time.perf_counter() - t0) / n
    out["pose_only_mse"] = float(r)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
