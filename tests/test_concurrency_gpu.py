"""GPU test (-m gpu): the tiled dense factorisation under oversubscription.

The reference runs its tracking, mapping and loop-closure bundlers concurrently with the per-frame ORB extraction and matching
(Tasks/Runtime.cpp:512-632: one worker per role, each with its own BundlerLib instances).  Here every handle owns a HIP stream and
the large-problem factorisation hands work between workgroups of one launch through bounded spins (chol_kernels.hip: split
diagonal tile, merged panel solve, persistent backward solve).  A stalled hand-off is reported as MAGE_ERR_DEVICE, never as a wrong
answer -- but under load it must not happen at all, and every handle must reproduce its solo run bit for bit."""
import threading

import numpy as np
import pytest

from mageslam_amd import frames, scene
from mageslam_amd.bundler import BundlerLib, load_scene
from mageslam_amd.orb import Matcher, OrbDetector

pytestmark = pytest.mark.gpu

N_BA, STEPS = 4, 50


def _ba_scene(i):
    # 152 cameras, 150 free: a 900 x 900 reduced system = 8 tiles of the tiled Cholesky (split diagonal, merged panel solves,
    # 8-stage backward solve); 6 000 points x 12 observations keeps a step at ~1 ms so that 50 steps of four handles overlap for real
    return scene.make_scene(n_cams=152, n_pts=6000, n_obs=72000, seed=0x5EED4000 + i, spacing=0.05, outlier_frac=0.01)


def _run_ba(s, steps, log):
    b = BundlerLib(False)
    load_scene(b, s, bulk=True)
    thr, out = 30.0, []
    for k in range(steps):
        mse = b.StepBundleAdjustment([1.8], thr, out)          # raises MageError on any status, MAGE_ERR_DEVICE included
        log.append((float(mse), len(out), b.trace()[-1]["trials"], b.trace()[-1]["code"]))
        if k % 10 == 9:
            thr *= 0.8
    return b.poses_f64(), b.points_f64(), list(out)


def test_four_tiled_solves_two_detectors_and_a_matcher_at_once():
    scenes = [_ba_scene(i) for i in range(N_BA)]
    assert BundlerLib is not None
    solo = []
    for s in scenes:
        log = []
        solo.append((_run_ba(s, STEPS, log), log))
    # the system really is tiled: order 900 -> 8 tiles
    b = BundlerLib(False); load_scene(b, scenes[0], bulk=True); b.StepBundleAdjustment([1.8], 1e30, [])
    z = b.structure("sizes")
    assert z[2] == 150 and z[8] == 1024, z
    b.close()

    fr = [frames.frame_pair(900 + i) for i in range(4)]
    det0 = OrbDetector()
    ref = [det0.DetectAndCompute(f[0]) for f in fr]
    ref_b = [det0.DetectAndCompute(f[1]) for f in fr]
    m0 = Matcher()
    ref_m = [m0.Match(ref[i][1], ref_b[i][1], None, None, 30, 1) for i in range(4)]

    errors, results = [], [None] * N_BA
    stop = threading.Event()
    gate = threading.Barrier(N_BA + 3)

    def ba(i):
        try:
            gate.wait()
            log = []
            results[i] = (_run_ba(scenes[i], STEPS, log), log)
        except BaseException as e:        # noqa: BLE001 - reported to the main thread
            errors.append(("ba", i, repr(e)))

    def orb(j):
        try:
            det = OrbDetector()
            gate.wait()
            n = 0
            while not stop.is_set():
                i = n % 4
                k, d = det.DetectAndCompute(fr[i][0])
                if not (np.array_equal(k["x"], ref[i][0]["x"]) and np.array_equal(k["y"], ref[i][0]["y"]) and np.array_equal(d, ref[i][1])):
                    raise AssertionError(f"detector {j}: frame {i} differs under load (iteration {n})")
                n += 1
        except BaseException as e:        # noqa: BLE001
            errors.append(("orb", j, repr(e)))

    def match():
        try:
            mt = Matcher()
            gate.wait()
            n = 0
            while not stop.is_set():
                i = n % 4
                if not np.array_equal(mt.Match(ref[i][1], ref_b[i][1], None, None, 30, 1), ref_m[i]):
                    raise AssertionError(f"matcher: pair {i} differs under load (iteration {n})")
                n += 1
        except BaseException as e:        # noqa: BLE001
            errors.append(("match", 0, repr(e)))

    th = [threading.Thread(target=ba, args=(i,)) for i in range(N_BA)]
    side = [threading.Thread(target=orb, args=(0,)), threading.Thread(target=orb, args=(1,)), threading.Thread(target=match)]
    for t in th + side:
        t.start()
    for t in th:
        t.join()
    stop.set()
    for t in side:
        t.join()
    assert not errors, errors
    for i in range(N_BA):
        (P, X, out), log = results[i]
        (P0, X0, out0), log0 = solo[i]
        assert log == log0, f"handle {i}: LM trace differs from its solo run"
        assert out == out0
        assert np.array_equal(P, P0) and np.array_equal(X, X0), f"handle {i}: state differs bitwise from its solo run"


def test_two_processes_oversubscribing_the_gpu_lose_no_step():
    """Several PROCESSES on one GPU (not how the library is deployed -- one process per GPU -- but what a shared development box
    does): the hardware scheduler time-slices them by saving and restoring workgroups, and a waiting panel-solve strip can then keep
    the producer it waits for off its compute unit until the bounded wait runs out.  The trial is re-run from the unchanged
    linearisation and the process switches to separate panel-solve launches (chol_report_stall): every step must succeed."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tools", "stall_probe.py"), "3", "25", tag], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for tag in ("A", "B")]
    for p in procs:
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, (out + err)[-2000:]
        res = json.loads([l for l in out.splitlines() if "{" in l][-1].split(" ", 1)[1])
        assert res["errors"] == [] and res["handles"] == 3 and res["steps"] == 25, res
