"""What a caller of the reference waits for: a bundler per optimisation (BundleAdjust.cpp:293, 348-351) -- create, Set*, ONE
StepBundleAdjustment (the reference's default local BA is one LM iteration, MageSettings.h:42-44), read the state back, destroy.

    python tools/one_shot_ba.py [--workload local] [--reps 50] [--steps 1]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def one_shot(s, steps=1, huber=0.9, thr=1e30, lam=None):
    from mageslam_amd.bundler import BundlerLib, load_scene
    t0 = time.perf_counter()
    b = BundlerLib(False)
    t1 = time.perf_counter()
    load_scene(b, s, bulk=True)
    if lam:
        b.SetCurrentLambda(lam)
    t2 = time.perf_counter()
    out = []
    b.StepBundleAdjustment([huber], thr, out)
    t3 = time.perf_counter()
    for _ in range(steps - 1):
        b.StepBundleAdjustment([huber], thr, out)
    t4 = time.perf_counter()
    b.GetPosesBulk(); b.GetPointsBulk()
    t5 = time.perf_counter()
    b.close()
    t6 = time.perf_counter()
    return dict(create=t1 - t0, set=t2 - t1, first_step=t3 - t2, more_steps=t4 - t3, get=t5 - t4, destroy=t6 - t5, total=t6 - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="local"); ap.add_argument("--reps", type=int, default=50); ap.add_argument("--steps", type=int, default=1)
    a = ap.parse_args()
    from mageslam_amd import scene
    s = scene.make_config(a.workload)
    for _ in range(3):
        one_shot(s, a.steps)
    rs = [one_shot(s, a.steps) for _ in range(a.reps)]
    print(json.dumps({"workload": a.workload, "steps": a.steps, "reps": a.reps,
                      **{k + "_ms": round(1e3 * float(np.median([r[k] for r in rs])), 4) for k in rs[0]}}))


if __name__ == "__main__":
    main()
