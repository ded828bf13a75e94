"""Aggregate LM-iteration rate of H independent 1k-pose sub-maps solved concurrently on ONE MI355X.

The reference runs several BundlerLib instances at once on different threads (mapping, loop closure, tracking).
Each handle owns its stream, so the chain-bound tail of one factorisation overlaps the MFMA bulk of another.  This is an
observation for DESIGN.md 5.1, not bench.py's headline (which stays one sub-map per GPU).

    python tools/bench_concurrent.py [--handles 2] [--steps 12] [--warmup 2]
"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--handles", type=int, default=2)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    from mageslam_amd import scene
    from mageslam_amd.bundler import BundlerLib, load_scene

    bs = []
    for i in range(a.handles):
        s = scene.make_scene(n_cams=1000, n_pts=100000, n_obs=1000000, seed=0x5EED0004 + 0x100 * i)
        b = BundlerLib(False)
        load_scene(b, s, bulk=True)
        for _ in range(a.warmup):
            b.StepBundleAdjustment([1.8], 1e30, [])
        bs.append(b)
    start = threading.Barrier(a.handles + 1)
    rmse = [0.0] * a.handles

    def work(i):
        start.wait()
        m = 0.0
        for _ in range(a.steps):
            m = bs[i].StepBundleAdjustment([1.8], 1e30, [])
        rmse[i] = float(m) ** 0.5

    th = [threading.Thread(target=work, args=(i,)) for i in range(a.handles)]
    for t in th: t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(json.dumps({"handles": a.handles, "steps_each": a.steps, "lm_iterations_per_s": a.handles * a.steps / dt,
                      "ms_per_iteration_per_handle": 1e3 * dt / a.steps, "rmse_px": rmse}))


if __name__ == "__main__":
    main()
