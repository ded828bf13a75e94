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
    s = scene.make_config("local")
    b = BundlerLib(False, device=0)
    load_scene(b, s, bulk=True)
    o = []
    for _ in range(3):
        b.StepBundleAdjustment([0.9], 1e30, o)
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        b.StepBundleAdjustment([0.9], 1e30, o)
    out["local_ms_per_lm_iteration"] = 1e3 * (time.perf_counter() - t0) / n
    out["local_rmse_px"] = float(np.sqrt(b.StepBundleAdjustment([0.9], 1e30, o)))
    b.close()
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
