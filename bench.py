#!/usr/bin/env python3
"""bench.py -- LM iterations/s of the MI355X bundle-adjustment back-end on BASELINE.json's headline workload.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N = 1 : the 1k-pose / 100k-point / 1M-observation synthetic global BA (BASELINE.json configs[3], the
          configuration the metric is quoted on) on cuda:0.
  N > 1 : launched by torch.distributed.run, one rank per GPU; every rank owns an independent sub-map of
          the same shape (different seed) -- BASELINE.json configs[4], weak scaling, no data-path collective.
A "step" is one Levenberg-Marquardt iteration = one BundlerLib::StepBundleAdjustment call with a single
Huber width (all its damped trials, plus the reference's outlier-classification pass).  Inputs are
resident in HBM when the timed region starts (the problem is uploaded and the graph structure is
built during warm-up).  Rank 0 prints ONE JSON line.

After the headline's timed region (never inside it) the same process measures, each bounded to seconds, what the other
BASELINE.json configurations and a caller of the reference see, and reports it under "extra" (N = 1) and "strong_scaling" (any N):
  extra.config2              ORB frames/s and matcher pairs/s at batch 1 / 64 / 1024 with the stage's HBM fraction, the BoW leaf lookup + IndexedMatch call, CPU oracle beside
  extra.pose_only / extra.reference_window   the reference's two EVERYDAY shapes (round 4), from native callers: the tracker's per-frame
                             pose refinement (1 free pose, 300 fixed points, 3 + 4 iterations, Tracking/TrackLocalMap.cpp:94-105, 421-501) and
                             the default local BA (12 keyframes / 400 points / 2000 observations, ONE iteration per bundler,
                             MageSettings.h:41-52, 77-78; BundleAdjust.cpp:281-354), create -> destroy phase by phase
                             (tools/shim_small_shapes.cpp over include/BundlerLib.h), the CPU oracle beside them from its own native
                             caller (oracle/small_shapes_main.c)
  extra.config3              local BA: steady ms per LM iteration AND create -> set -> one iteration -> read back -> destroy (the
                             reference builds a bundler per optimisation, BundleAdjust.cpp:293, 348-351; its default local BA is
                             ONE iteration, MageSettings.h:42-44), the reference's 10-call caller loop, CPU oracle beside
  extra.config4_end_to_end   loop-closure global BA as Console configures it (25 iterations in one call, console.cpp:115-120)
                             including create / set / structure build / read back / destroy, g2o's own lambda start
  extra.sustained            >= 2 s of headline steps (scene re-loaded every 20 so that they stay single-trial), per-block spread;
                             the same from g2o's own lambda start; one handle stepped 120 times without re-loading
  extra.concurrent_handles   1 / 2 / 4 sub-maps stepped concurrently on this GPU (aggregate LM iterations/s: a labelled secondary)
  extra.config5_windowed_1gpu / strong_scaling   ONE 8k-pose map in 8 keyframe windows, 8 / N windows per rank, pose block
                             all-reduced in HBM (RCCL when N > 1): outer iterations/s -- the strong-scaling curve's point for this N
--no-extras skips them; --submaps-per-gpu K steps K sub-maps per GPU concurrently in the replica mode (a labelled secondary).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E (/opt/skills/guides/MI355X_MICROARCH.md: ~8 TB/s peak, ~6.3 TB/s achievable)
F64_MFMA_PEAK_TFLOPS = 78.6   # v_mfma_f64_16x16x4_f64: 32 flop/clk/SIMD * 4 SIMD * 256 CU * 2.4 GHz (AMD MI355X spec)
HUBER = 1.8                   # BundleAdjustSettings.HuberWidth default (MageSettings.h:41-52)
# "for throughput runs seed lambda so that >= 95 % of iterations take exactly 1 trial" (report trials/iter).  With
# g2o's own start (tau * max diag = 500) the damping has decayed to ~2e-4 by iteration 19 and a handful of iterations then
# need a second trial; SetCurrentLambda(5e6) (BundlerLib.cpp:354-362) moves that phase past iteration 26.  Same minimum
# (RMSE 1.30294 px) either way; a second trial only ever ADDS work to an iteration.
LAMBDA_SEED = {"global": 5e6}
WORKLOADS = {
    "global": dict(n_cams=1000, n_pts=100000, n_obs=1000000, seed=0x5EED0004),
    "local": dict(n_cams=20, n_pts=5000, n_obs=50000, seed=0x5EED0003, fixed=(0, 1, 15, 16, 17, 18, 19)),
    "tiny": dict(n_cams=10, n_pts=200, n_obs=2000, seed=0x5EED0001),
}


def _oracle_iterations(args):
    """One replica of the CPU baseline: n LM iterations of the oracle on the workload's scene (seed offset per replica)."""
    workload, n_iter, replica = args
    import time as _t
    from mageslam_amd import scene
    from oracle.oracle import OracleBundler, load_scene_bulk
    kw = dict(WORKLOADS[workload])
    kw["seed"] = kw["seed"] + 0x100 * replica
    s = scene.make_scene(**kw)
    b = OracleBundler()
    load_scene_bulk(b, s)
    if workload in LAMBDA_SEED:
        b.SetCurrentLambda(LAMBDA_SEED[workload])
    out: list = []
    t0 = _t.perf_counter()
    for _ in range(n_iter):
        b.StepBundleAdjustment([HUBER], 1e30, out)
    return _t.perf_counter() - t0


def cpu_baseline(workload: str, min_iterations: int = 2, max_seconds: float = 45.0, replicas: int = 8) -> dict:
    """Times the CPU oracle (oracle/ba_oracle.c: same algorithm incl. the reference's unblocked pivoted LDLT, one thread like the
    reference's single-threaded g2o path) on the same scene for a bounded sample: at least `min_iterations` LM iterations on
    one core, then `replicas` independent sub-maps side by side, one per core (the multi-GPU comparison)."""
    el1 = _oracle_iterations((workload, 1, 0))
    n = max(min_iterations, min(50, int(10.0 / max(el1, 1e-9))))
    if n * el1 > max_seconds:
        n = max(min_iterations, int(max_seconds / el1))
    el = _oracle_iterations((workload, n, 0))
    res = {"value": n / el, "unit": "LM iterations/s", "cores": 1, "host_cores": os.cpu_count(), "kind": "port",
           "sample": f"{n} LM iteration(s) of the same {workload} scene from the same initial state, {el:.1f} s, "
                     f"oracle/ba_oracle.c (gcc -O3), 1 thread"}
    try:
        import multiprocessing as mp
        r = min(replicas, os.cpu_count() or 1)
        t0 = time.perf_counter()
        with mp.get_context("spawn").Pool(r) as pool:
            pool.map(_oracle_iterations, [(workload, 1, i) for i in range(r)])
        wall = time.perf_counter() - t0
        res["replicas"] = {"value": r / wall, "unit": "LM iterations/s", "cores": r,
                           "sample": f"{r} independent sub-maps, one oracle process per core, 1 LM iteration each, {wall:.1f} s wall "
                                     f"(scene generation included)"}
    except Exception as e:  # noqa: a report only
        res["replicas"] = {"value": None, "sample": f"failed: {e}"}
    return res



# ======================================================================================================================
# extras: measured after the headline's timed region, each bounded to seconds; a failure is recorded, never raised
# ======================================================================================================================
def _latest_profile(suffix: str) -> str:
    """profiles/r<NN>_<suffix> of the latest round that has one (one copy per round, no unprefixed duplicates)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    if not files:
        raise FileNotFoundError(suffix)
    return files[-1]


def _finite(x):
    """JSON has no NaN / Infinity: non-finite numbers become null."""
    if isinstance(x, float):
        return x if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    return x


def _median_ms(xs):
    return round(1e3 * float(np.median(xs)), 4)


def _read_back(b):
    """What the caller does after an optimisation: every pose and every map point (BundleAdjust.cpp:318-347)."""
    b.GetPosesBulk()
    (b.GetPointsBulk if hasattr(b, "GetPointsBulk") else b.points_f64)()


def _one_shot(make, load, s, n_steps_per_call, hubers, thr, lam=None, calls=1):
    """create -> Set* -> `calls` x StepBundleAdjustment(hubers) -> read the state back -> destroy; returns seconds per phase."""
    t0 = time.perf_counter()
    b = make()
    t1 = time.perf_counter()
    load(b, s)
    if lam:
        b.SetCurrentLambda(lam)
    t2 = time.perf_counter()
    out: list = []
    mse = b.StepBundleAdjustment(hubers, thr, out)
    t3 = time.perf_counter()
    for _ in range(calls - 1):
        mse = b.StepBundleAdjustment(hubers, thr, out)
    t4 = time.perf_counter()
    _read_back(b)
    t5 = time.perf_counter()
    b.close()
    t6 = time.perf_counter()
    return dict(create=t1 - t0, set=t2 - t1, first_call=t3 - t2, more_calls=t4 - t3, get=t5 - t4, destroy=t6 - t5, total=t6 - t0, mse=float(mse))


def _caller_loop(make, load, s, huber=0.9):
    """The reference's local-BA caller (BA-16, BundleAdjust.cpp:281-354): a bundler per run, 10 calls of one iteration each with
    the outlier threshold shrinking by MaxOutlierErrorScaleFactor^2 per call, then the state read back."""
    t0 = time.perf_counter()
    b = make()
    load(b, s)
    thr, out = 7.25, []
    for _ in range(10):
        b.StepBundleAdjustment([huber], thr, out)
        thr *= 0.95 * 0.95
    _read_back(b)
    b.close()
    return time.perf_counter() - t0, len(out)


def _cpp_caller_steady(s):
    """The same one-iteration call from the reference's own kind of caller: tools/shim_local_ba.cpp (C++ against include/BundlerLib.h,
    built by __graft_entry__.build_tools) -- median and minimum of 8 calls; None when the tool is not built."""
    import subprocess
    import tempfile
    from mageslam_amd import scene
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "_bin", "shim_local_ba")
    if not os.path.exists(exe):
        return None
    try:
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "local.scene")
            scene.save_scene(s, path)
            out = subprocess.run([exe, path, "0.9", "--steady", "8"], capture_output=True, text=True, timeout=120).stdout
        for line in out.splitlines():
            if line.startswith("steady_ms"):
                med, mn = line.split()[1:3]
                return {"median_ms": round(float(med), 4), "min_ms": round(float(mn), 4), "calls": 8}
    except Exception as e:                                  # a missing compiler run-time on the box must not cost the bench line
        return {"error": str(e)[:200]}
    return None



# ---- the reference's everyday shapes, both sides from native callers -------------------------------------------------------------
SMALL_SHAPES = {
    # Tracking/TrackLocalMap.cpp:421-501: one free pose, the frame's ~300 associated map points (fixed, accurate to a few mm), a pose
    # prior off by a couple of centimetres, 5 % wrong associations
    "pose_only": dict(mode="pose-only", scene=dict(n_cams=1, n_pts=300, n_obs=300, seed=31, fixed=(), outlier_frac=0.05, pt_sigma=0.004, cam_sigma=0.02, rot_sigma=0.005),
                      what="TrackLocalMap::OptimizeCameraPose twice per frame: 1 free pose / 300 fixed points; pass 1 = 3 iterations, Huber 4.0, 36.0; "
                           "pass 2 = 4 iterations, Huber 0.9, 20.25 (MageSettings.h:180-195); make_unique<BundlerLib> -> Set* per element -> one "
                           "StepBundleAdjustment -> GetPose -> reset"),
    # MageSettings.h:77-78 (1500-2000 connections), :41-52 (NumSteps = NumStepsPerRun = 1, Huber 1.8, MaxOutlierError 7.25 passed un-squared)
    "reference_window": dict(mode="window", scene=dict(n_cams=12, n_pts=400, n_obs=2000, seed=0x5EED0012, fixed=(0, 1, 2, 3), outlier_frac=0.02),
                             what="BundleAdjust::RunBundleAdjustment with the default settings: 12 keyframes (4 fixed) / 400 points / 2000 observations, "
                                  "MakeBundler -> BuildDataForG2O (per element) -> ONE StepBundleAdjustment({1.8}, 7.25) -> UpdateData (GetPose / GetPoint per "
                                  "element) -> reset"),
}


def _small_shape_lines(text: str) -> dict:
    """Parses the lines both native callers print: '<tag> total T min M create C set S step P get G destroy D calls N' and the result lines."""
    out = {}
    for line in text.splitlines():
        tok = line.split()
        if len(tok) >= 3 and tok[1] == "total":
            out[tok[0]] = {tok[i]: (int(tok[i + 1]) if tok[i] == "calls" else round(float(tok[i + 1]), 5)) for i in range(1, len(tok) - 1, 2)}
        elif tok and tok[0] in ("pose_only_frame_ms", "window_result"):
            out[tok[0]] = " ".join(tok[1:])
    return out


def _run_small_shapes(exe: str, reps: int = 300) -> dict:
    import subprocess
    import tempfile
    from mageslam_amd import scene
    res = {}
    if not os.path.exists(exe):
        return {"error": f"{os.path.relpath(exe, ROOT)} is not built"}
    with tempfile.TemporaryDirectory() as d:
        for name, cfg in SMALL_SHAPES.items():
            path = os.path.join(d, name + ".scene")
            scene.save_scene(scene.make_scene(**cfg["scene"]), path)
            try:
                txt = subprocess.run([exe, cfg["mode"], path, str(reps)], capture_output=True, text=True, timeout=180).stdout
                res[name] = _small_shape_lines(txt)
            except Exception as e:  # noqa: BLE001 - a report only
                res[name] = {"error": str(e)[:200]}
    return res


def extra_small_shapes() -> dict:
    """Device path: tools/_bin/shim_small_shapes (C++ over include/BundlerLib.h, the reference's call protocol, per-element setters)."""
    r = _run_small_shapes(os.path.join(ROOT, "tools", "_bin", "shim_small_shapes"))
    for name, cfg in SMALL_SHAPES.items():
        if isinstance(r.get(name), dict):
            r[name]["workload"] = cfg["what"]
            r[name]["caller"] = "tools/shim_small_shapes.cpp (C++, include/BundlerLib.h); medians in ms over the calls, `min` = fastest call"
    return r


def cpu_baseline_small_shapes() -> dict:
    """CPU oracle on the same scenes from ITS native caller (oracle/small_shapes_main.c, gcc -O3, one thread like the reference's g2o path)."""
    r = _run_small_shapes(os.path.join(ROOT, "oracle", "_build", "oracle_small_shapes"))
    r["kind"] = "port"; r["cores"] = 1
    return r

def extra_config3(device):
    from mageslam_amd import scene
    from mageslam_amd.bundler import BundlerLib, load_scene
    s = scene.make_scene(**WORKLOADS["local"])
    so = scene.make_scene(**WORKLOADS["local"], outlier_frac=0.02)
    make = lambda: BundlerLib(False, device=device)
    load = lambda b, sc: load_scene(b, sc, bulk=True)
    for _ in range(3):
        _one_shot(make, load, s, 1, [0.9], 1e30)
    shots = [_one_shot(make, load, s, 1, [0.9], 1e30) for _ in range(40)]
    one = {k + "_ms": _median_ms([r[k] for r in shots]) for k in ("create", "set", "first_call", "get", "destroy", "total")}
    # steady state: structure built, nothing removed, no re-seed of lambda
    b = make(); load(b, s)
    for _ in range(2):
        b.StepBundleAdjustment([0.9], 1e30, [])
    t0 = time.perf_counter()
    n = 8
    for _ in range(n):
        b.StepBundleAdjustment([0.9], 1e30, [])
    steady = (time.perf_counter() - t0) / n
    trials = [t["trials"] for t in b.trace()]
    b.close()
    cpp = _cpp_caller_steady(s)
    for _ in range(2):
        _caller_loop(make, load, so)
    loops = [_caller_loop(make, load, so) for _ in range(20)]
    return {"workload": "local: 20 keyframes / 5000 points / 50000 observations, Huber 0.9, 7 keyframes fixed",
            "steady_ms_per_lm_iteration": round(1e3 * steady, 4), "steady_last_trials": trials,
            "steady_ms_per_lm_iteration_cpp_caller": cpp,
            "one_iteration_bundler_create_to_destroy": one,
            "caller_loop_10_calls_ms": _median_ms([t for t, _ in loops]), "caller_loop_outliers_removed": loops[-1][1],
            "structure_build": "device (mageslam_amd/csrc/ba_build.hip)" if os.environ.get("MAGE_BA_BUILD", "d")[0] == "d" else "host"}


def cpu_baseline_config3():
    """The CPU oracle on the same local-BA legs (one thread, like the reference's g2o path)."""
    from mageslam_amd import scene
    from oracle.oracle import OracleBundler, load_scene_bulk
    s = scene.make_scene(**WORKLOADS["local"])
    so = scene.make_scene(**WORKLOADS["local"], outlier_frac=0.02)
    make = lambda: OracleBundler(False)
    shots = [_one_shot(make, load_scene_bulk, s, 1, [0.9], 1e30) for _ in range(5)]
    b = make(); load_scene_bulk(b, s)
    b.StepBundleAdjustment([0.9], 1e30, [])
    t0 = time.perf_counter()
    for _ in range(5):
        b.StepBundleAdjustment([0.9], 1e30, [])
    steady = (time.perf_counter() - t0) / 5
    loops = [_caller_loop(make, load_scene_bulk, so) for _ in range(3)]
    return {"kind": "port", "cores": 1, "steady_ms_per_lm_iteration": round(1e3 * steady, 3),
            "one_iteration_bundler_create_to_destroy_ms": _median_ms([r["total"] for r in shots]),
            "caller_loop_10_calls_ms": _median_ms([t for t, _ in loops])}


def extra_config4_end_to_end(device):
    from mageslam_amd import scene
    from mageslam_amd.bundler import BundlerLib, load_scene
    s = scene.make_scene(**WORKLOADS["global"])
    make = lambda: BundlerLib(False, device=device)
    load = lambda b, sc: load_scene(b, sc, bulk=True)
    _one_shot(make, load, s, 25, [0.372231] * 25, 7.25)
    runs = [_one_shot(make, load, s, 25, [0.372231] * 25, 7.25) for _ in range(3)]
    first = [_one_shot(make, load, s, 1, [HUBER], 1e30) for _ in range(5)]
    return {"workload": "global: 1000 / 100000 / 1000000, ONE StepBundleAdjustment call of 25 iterations, Huber 0.372231, "
                        "MaxOutlierError 7.25 (console.cpp:115-120), g2o's own lambda start, create -> destroy",
            "total_ms": _median_ms([r["total"] for r in runs]), "set_ms": _median_ms([r["set"] for r in runs]),
            "step_call_ms": _median_ms([r["first_call"] for r in runs]), "get_ms": _median_ms([r["get"] for r in runs]),
            "rmse_px": round(float(np.sqrt(runs[-1]["mse"])), 5),
            "one_iteration_bundler": {"first_step_ms_structure_build_plus_one_iteration": _median_ms([r["first_call"] for r in first]),
                                      "create_to_destroy_ms": _median_ms([r["total"] for r in first])}}


def extra_cold_start(device):
    """The first bundle adjustment of a new map size in a FRESH process (tools/cold_start.py), at the headline size and at 2 000 poses."""
    import subprocess
    out = {}
    for wl, then in (("global", "global1200"), ("global2k", "")):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cold_start.py"), "--workload", wl, "--device", str(device)] + (["--then", then] if then else []),
                           capture_output=True, text=True, timeout=600)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        out[wl] = json.loads(lines[-1]) if p.returncode == 0 and lines else {"error": (p.stdout + p.stderr)[-400:]}
    return out


def extra_skyline_solve(device, steps=20):
    """LABELLED SECONDARY: the same map with mage_ba_use_skyline -- the dense solve's schedule skips the tiles left of the reduced system's
    skyline (the benchmark scene's S is block-banded; the headline treats it as dense, as the reference's LinearSolverDense does, SURVEY
    8d).  Same numbers to the bit; the factorisation is then bound by its chain of diagonal tiles alone."""
    import subprocess
    from mageslam_amd import scene
    from mageslam_amd.bundler import BundlerLib, load_scene
    s = scene.make_scene(**WORKLOADS["global"])
    out = {}
    mses = {}
    for tag, sky in (("dense", False), ("skyline", True)):
        b = BundlerLib(False, device=device)
        b.use_skyline(sky)
        load_scene(b, s, bulk=True)
        b.SetCurrentLambda(LAMBDA_SEED["global"])
        o: list = []
        for _ in range(4):          # (as the headline: a few warm steps, then 20 timed ones -- beyond ~30 iterations the map has converged and trials are rejected)
            b.StepBundleAdjustment([HUBER], 1e30, o)
        b.enable_profiling(2)
        t0 = time.perf_counter()
        for _ in range(steps):
            mse = b.StepBundleAdjustment([HUBER], 1e30, o)
        dt = time.perf_counter() - t0
        p = b.profile()
        out[tag] = {"lm_iterations_per_s": round(steps / dt, 2), "ms_per_step": round(1e3 * dt / steps, 4),
                    "factor_and_solves_ms": round(p.factor_ms_total / max(int(p.n_factorizations), 1), 4)}
        mses[tag] = float(mse)
        b.close()
    out["same_mse_to_the_bit"] = mses["dense"] == mses["skyline"]
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cold_start.py"), "--workload", "global2k", "--device", str(device), "--skyline"],
                        capture_output=True, text=True, timeout=600)
    lines = [l for l in pr.stdout.splitlines() if l.startswith("{")]
    out["global2k_skyline"] = json.loads(lines[-1])["fresh_process"] if pr.returncode == 0 and lines else {"error": (pr.stdout + pr.stderr)[-400:]}
    out["note"] = "LABELLED SECONDARY (mage_ba_use_skyline / MAGE_BA_SKYLINE=1): not the headline, whose solve is dense; roofline counts n^3/3 only there"
    return out


def extra_sustained(device, min_seconds=2.0):
    from mageslam_amd import scene
    from mageslam_amd.bundler import BundlerLib, load_scene
    s = scene.make_scene(**WORKLOADS["global"])

    def blocks(lam, seconds):
        per_block, trials, steps = [], 0, 0
        t_all = time.perf_counter()
        while sum(per_block) < seconds and time.perf_counter() - t_all < 4 * seconds + 5:
            b = BundlerLib(False, device=device)
            load_scene(b, s, bulk=True)
            if lam:
                b.SetCurrentLambda(lam)
            out: list = []
            for _ in range(3):
                b.StepBundleAdjustment([HUBER], 1e30, out)
            t0 = time.perf_counter()
            for _ in range(20):
                b.StepBundleAdjustment([HUBER], 1e30, out)
                trials += sum(t["trials"] for t in b.trace())
            per_block.append(time.perf_counter() - t0)
            steps += 20
            b.close()
        ms = [1e3 * t / 20 for t in per_block]
        return {"steps": steps, "timed_seconds": round(sum(per_block), 3), "wall_seconds": round(time.perf_counter() - t_all, 3),
                "ms_per_step": round(float(np.mean(ms)), 4), "ms_per_step_first_block": round(ms[0], 4), "ms_per_step_last_block": round(ms[-1], 4),
                "ms_per_step_min_block": round(min(ms), 4), "ms_per_step_max_block": round(max(ms), 4),
                "lm_iterations_per_s": round(1e3 / float(np.mean(ms)), 2), "trials_per_iteration": round(trials / steps, 4)}

    res = {"seeded_lambda_5e6": blocks(LAMBDA_SEED["global"], min_seconds), "g2o_default_lambda": blocks(None, min_seconds / 2)}
    # one handle, no re-load: the map converges after ~30 iterations and LM then takes extra damped trials per iteration
    b = BundlerLib(False, device=device)
    load_scene(b, s, bulk=True)
    b.SetCurrentLambda(LAMBDA_SEED["global"])
    out: list = []
    tr = []
    t0 = time.perf_counter()
    n_steps = 120
    for _ in range(n_steps):
        b.StepBundleAdjustment([HUBER], 1e30, out)
        tr.append(b.trace()[-1]["trials"])
    el = time.perf_counter() - t0
    res["one_handle_120_steps_no_reload"] = {"seconds": round(el, 3), "ms_per_step": round(1e3 * el / n_steps, 4),
                                             "trials_per_iteration": round(float(np.mean(tr)), 4),
                                             "ms_per_trial": round(1e3 * el / max(sum(tr), 1), 4),
                                             "trials_first_30": round(float(np.mean(tr[:30])), 3), "trials_last_30": round(float(np.mean(tr[-30:])), 3),
                                             "note": "the map has converged after ~30 iterations; every later iteration then ends in g2o's 10 rejected trials (Terminate), "
                                                     "so ms_per_trial, not ms_per_step, is the comparable figure"}
    b.close()
    return res


def extra_concurrent_handles(device, counts=(1, 2, 4), steps=12):
    import threading
    from mageslam_amd import scene
    from mageslam_amd.bundler import BundlerLib, load_scene
    res = {}
    bs = []
    for i in range(max(counts)):
        s = scene.make_scene(**dict(WORKLOADS["global"], seed=WORKLOADS["global"]["seed"] + 0x100 * i))
        b = BundlerLib(False, device=device)
        load_scene(b, s, bulk=True)
        b.SetCurrentLambda(LAMBDA_SEED["global"])
        for _ in range(2):
            b.StepBundleAdjustment([HUBER], 1e30, [])
        bs.append(b)
    for hn in counts:
        gate = threading.Barrier(hn + 1)

        def work(i):
            gate.wait()
            for _ in range(steps):
                bs[i].StepBundleAdjustment([HUBER], 1e30, [])
        th = [threading.Thread(target=work, args=(i,)) for i in range(hn)]
        for t in th:
            t.start()
        gate.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        res[str(hn)] = {"lm_iterations_per_s_aggregate": round(hn * steps / dt, 2), "ms_per_iteration_per_handle": round(1e3 * dt / steps, 4)}
    for b in bs:
        b.close()
    res["note"] = "independent 1k-pose sub-maps on concurrent host threads, one stream each; the headline stays ONE sub-map per GPU"
    return res


def _bow_workload(cap=440, kk=10, depth=4, seed=5):
    """A 10-ary vocabulary tree of depth 4 (11 111 nodes, random medoids) and two frames' descriptors: the shape of a relocalisation query."""
    rng = np.random.default_rng(seed)
    n_nodes = sum(kk ** d for d in range(depth + 1))
    inner = sum(kk ** d for d in range(depth))
    node_desc = rng.integers(0, 256, (n_nodes, 32), dtype=np.uint8)
    child_off = np.zeros(n_nodes + 1, np.int32)
    child_off[1:inner + 1] = kk * np.arange(1, inner + 1)
    child_off[inner + 1:] = child_off[inner]
    children = np.arange(1, n_nodes, dtype=np.int32)
    A = rng.integers(0, 256, (cap, 32), dtype=np.uint8)
    B = A.copy(); B[:, 0] ^= rng.integers(0, 4, cap, dtype=np.uint8)
    return node_desc, child_off, children, A, B


def _leaf_csr(leaves, n_nodes):
    order = np.argsort(leaves, kind="stable").astype(np.int32)
    off = np.zeros(n_nodes + 1, np.int32)
    np.add.at(off, leaves + 1, 1)
    return np.cumsum(off).astype(np.int32), order


def extra_config2(device):
    import threading
    import torch
    from mageslam_amd import frames
    from mageslam_amd.orb import Matcher, OrbDetector
    W, H, CAP = 640, 480, 440
    base = [frames.frame_pair(500 + i) for i in range(8)]
    a_set = np.stack([p[0] for p in base]); b_set = np.stack([p[1] for p in base])
    det, mt = OrbDetector(device=device), Matcher(device=device)
    res = {"workload": "640x480 synthetic frame pairs, default extractor settings (440 keypoints), MaxHammingDistance 30 / MinHammingDifference 1",
           "batches": {}}
    for batch in (1, 64, 1024):
        imgs = torch.from_numpy(np.concatenate([a_set, b_set])[np.arange(2 * batch) % 16]).to(f"cuda:{device}").contiguous()
        torch.cuda.synchronize()
        for _ in range(2):
            kp, de, cn = det.detect_batch_device(imgs.data_ptr(), 2 * batch, W, H, CAP)
        reps = 20 if batch < 1024 else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            kp, de, cn = det.detect_batch_device(imgs.data_ptr(), 2 * batch, W, H, CAP)
        wall = (time.perf_counter() - t0) / reps
        det.enable_profile(True)
        for _ in range(2):
            kp, de, cn = det.detect_batch_device(imgs.data_ptr(), 2 * batch, W, H, CAP)
        p = det.profile()
        det.enable_profile(False)
        fps2 = None
        if batch >= 64:        # two detectors fed from two host threads (the reference's ImageAnalyzer runs one per camera thread)
            dets = [det, OrbDetector(device=device)]
            for d in dets:
                d.detect_batch_device(imgs.data_ptr(), 2 * batch, W, H, CAP)
            gate = threading.Barrier(3)

            def loop(d):
                gate.wait()
                for _ in range(reps):
                    d.detect_batch_device(imgs.data_ptr(), 2 * batch, W, H, CAP)
                gate.wait()
            th = [threading.Thread(target=loop, args=(d,)) for d in dets]
            for t in th:
                t.start()
            gate.wait(); t0 = time.perf_counter(); gate.wait()
            fps2 = 2 * reps * 2 * batch / (time.perf_counter() - t0)
            for t in th:
                t.join()
            kp, de, cn = det.detect_batch_device(imgs.data_ptr(), 2 * batch, W, H, CAP)
        alg_bytes = 2 * batch * (W * H * 3)      # FAST read 0.31 MB + blur read and write 0.61 MB = 0.92 MB / frame (the 0.23 MB of BRIEF gathers are L2-cached re-reads of the blurred frame, the 26 KB of output are noise)
        dA, cA = de, cn
        dB, cB = de + batch * CAP * 32, cn + batch * 4
        for _ in range(2):
            mt.match_batch_device(batch, dA, cA, CAP, dB, cB, CAP, 30, 1)
        t0 = time.perf_counter()
        for _ in range(reps):
            mt.match_batch_device(batch, dA, cA, CAP, dB, cB, CAP, 30, 1)
        mwall = (time.perf_counter() - t0) / reps
        # `valu_issue` rooflines (the bound the front end is really under: SURVEY 8d's HBM figure says little about kernels that are
        # limited by vector-instruction issue).  Per kernel: wavefronts of the launch x vector instructions per wavefront (PMC pass,
        # profiles/front_end_valu.json "pmc") x mean issue cycles per instruction (static mix by issue class, "mix") = SIMD-cycles of issue
        # the launch NEEDS, against the SIMD-cycles it HAD: 1024 SIMDs x its measured time x 2.4 GHz.
        valu = {}
        try:
            fe = json.load(open(os.path.join(ROOT, "profiles", "front_end_valu.json")))
            SIMD_HZ = 1024 * 2.4e9
            for kern, units, ms in (("k_fast_keypoints", 2 * batch, p.fast_ms + p.blur_ms), ("k_select", 2 * batch, p.select_ms), ("k_brief", 2 * batch, p.brief_ms),
                                    ("k_match", batch, mt.last_kernel_ms())):
                pm, mx = fe["pmc"][kern], fe["mix"][kern]
                per_unit = pm["waves_per_launch"] / pm["units_per_launch"]
                instr = units * per_unit * pm["valu_per_wave"]                 # vector wave-instructions of the launch
                FAST = 2.28                                                    # cycles: the shortest issue interval measured on this part (profiles/r03_valu_issue_rates.txt)
                need_lo, need_hi = instr * FAST, instr * mx["mean_issue_cycles_per_valu"]
                valu[kern] = {"bound": "valu_issue", "achieved": round(instr / (ms * 1e-3) / 1e12, 3), "peak": round(SIMD_HZ / FAST / 1e12, 3), "unit": "T vector wave-instructions/s",
                              "frac": round(need_lo / (ms * 1e-3) / SIMD_HZ, 4),
                              "frac_if_classes_serialise": round(need_hi / (ms * 1e-3) / SIMD_HZ, 4),
                              "ms": round(ms, 4), "wavefronts": int(units * per_unit), "valu_instructions_per_wavefront": round(pm["valu_per_wave"], 1),
                              "static_mix_mean_issue_cycles": mx["mean_issue_cycles_per_valu"]}
            valu["how_to_read"] = ("frac = vector wave-instructions x 2.28 cycles (the shortest measured issue interval) / (1024 SIMDs x 2.4 GHz x the launch's time): "
                                   "the share of issue slots taken, a LOWER bound of how busy the vector unit is; frac_if_classes_serialise weights the kernel's STATIC "
                                   "instruction mix with the per-class intervals (2.3 / 4.2 / 8.2 / 16 cycles) as if classes never overlapped -- an upper estimate "
                                   "(it can exceed 1: the 4.2-cycle classes do overlap with other wavefronts' 2.3-cycle ones)")
            valu["source"] = "instruction counts: committed PMC pass (profiles/front_end_valu.json, tools/pmc_front_end.sh) -- NOT measured in this run; times: this run"
        except Exception as e:  # noqa: BLE001 - the profile is optional
            valu = {"error": f"{type(e).__name__}: {e}"}
        res["batches"][str(batch)] = {
            "frames_per_s": round(2 * batch / wall, 1), "frames_per_s_two_detectors": None if fps2 is None else round(fps2, 1),
            "ms_per_batch_of_frames": round(1e3 * wall, 4), "frames_in_batch": 2 * batch,
            "stage_ms": {"fast_blur": round(p.fast_ms + p.blur_ms, 4), "select": round(p.select_ms, 4), "brief": round(p.brief_ms, 4), "events_total": round(p.total_ms, 4)},
            "hbm_frac_algorithmic": round(alg_bytes / (p.total_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4),
            "roofline_valu_issue": valu,
            "pairs_per_s": round(batch / mwall, 1), "match_kernel_ms": round(mt.last_kernel_ms(), 4),
            "match_gdistances_per_s": round(batch * 2 * CAP * CAP / (mt.last_kernel_ms() * 1e-3) / 1e9, 2)}
    # the vocabulary's leaf lookup and IndexedMatch through it, host buffers in and out, the tree upload included
    node_desc, child_off, children, A, B = _bow_workload(CAP)
    both = np.concatenate([A, B])
    leaves = mt.BowFindLeaf(node_desc, child_off, children, both)
    fao, fa = _leaf_csr(leaves[:CAP], len(node_desc)); fbo, fb = _leaf_csr(leaves[CAP:], len(node_desc))
    got = mt.IndexedMatchBow(node_desc, child_off, children, A, fao, fa, B, fbo, fb, 30, 1)
    reps = 100
    t0 = time.perf_counter()
    for _ in range(reps):
        mt.BowFindLeaf(node_desc, child_off, children, both)
    leaf_ms = 1e3 * (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        mt.IndexedMatchBow(node_desc, child_off, children, A, fao, fa, B, fbo, fb, 30, 1)
    im_ms = 1e3 * (time.perf_counter() - t0) / reps
    mt.BowSetTree(node_desc, child_off, children)            # the tree kept on the device (it changes once, when its training completes)
    assert np.array_equal(mt.BowFindLeaf(None, None, None, both), leaves)
    t0 = time.perf_counter()
    for _ in range(reps):
        mt.BowFindLeaf(None, None, None, both)
    leaf_kept_ms = 1e3 * (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        mt.IndexedMatchBow(None, None, None, A, fao, fa, B, fbo, fb, 30, 1)
    im_kept_ms = 1e3 * (time.perf_counter() - t0) / reps
    mt.BowSetTree()
    res["bow"] = {"workload": f"10-ary vocabulary tree of depth 4 ({len(node_desc)} nodes), {2 * CAP} descriptors per call, host buffers in and out (tree upload included)",
                  "find_leaf_ms_per_call": round(leaf_ms, 4), "indexed_match_bow_ms_per_call": round(im_ms, 4),
                  "find_leaf_tree_kept_on_device_ms_per_call": round(leaf_kept_ms, 4), "indexed_match_bow_tree_kept_on_device_ms_per_call": round(im_kept_ms, 4), "matches": int(len(got)),
                  "leaf_checksum": int(np.asarray(leaves, np.int64).sum())}
    return res


def cpu_baseline_config2(seconds=2.0):
    """The CPU oracle on the same frames (one thread; the reference runs OpenCV single-threaded, cv::setNumThreads(0))."""
    from mageslam_amd import frames
    from oracle import oracle as O
    base = [frames.frame_pair(500 + i) for i in range(4)]
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        O.orb_detect(base[n % 4][0]); n += 1
    fps = n / (time.perf_counter() - t0)
    ka, da = O.orb_detect(base[0][0]); kb, db = O.orb_detect(base[0][1])
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        O.match(da, db, 30, 1); n += 1
    pps = n / (time.perf_counter() - t0)
    node_desc, child_off, children, A, B = _bow_workload()
    both = np.concatenate([A, B])
    t0 = time.perf_counter(); leaves = O.bow_find_leaf(node_desc, child_off, children, both); leaf_ms = 1e3 * (time.perf_counter() - t0)
    fao, fa = _leaf_csr(leaves[:len(A)], len(node_desc)); fbo, fb = _leaf_csr(leaves[len(A):], len(node_desc))
    t0 = time.perf_counter(); O.indexed_match_bow(node_desc, child_off, children, A, fao, fa, B, fbo, fb, 30, 1); im_ms = 1e3 * (time.perf_counter() - t0)
    return {"kind": "port", "cores": 1, "frames_per_s": round(fps, 1), "pairs_per_s": round(pps, 1),
            "bow_find_leaf_ms_per_call": round(leaf_ms, 4), "bow_indexed_match_ms_per_call": round(im_ms, 4), "bow_leaf_checksum": int(np.asarray(leaves, np.int64).sum()),
            "sample": f"{seconds:.0f} s of oracle/orb_oracle.c on 640x480 frames + {seconds:.0f} s of oracle/match_oracle.c on one 440 x 440 pair + one BoW lookup / IndexedMatch call of extra.config2.bow"}


def strong_scaling_windowed(device, dist, rank, world, poses=8000, windows=8, overlap=10, iters=10, warmup=2, threads=4):
    """BASELINE.json configs[4], second form: ONE 8k-pose map cut into 8 keyframe windows, 8 / N windows per rank; an outer
    iteration = one LM iteration in every window + the all-reduce of the (poses x 8) f64 pose block in HBM (RCCL when N > 1)."""
    import torch
    from mageslam_amd import dist as D, scene
    from mageslam_amd.bundler import BundlerLib, load_scene
    from mageslam_amd.windowed import WindowedMap
    t0 = time.perf_counter()
    # Everything up to here is local to a rank.  The ranks agree that ALL of them have their windows before the first collective of
    # the leg: a rank that failed to build (out of memory, ...) must not leave the others in an all-reduce.
    m, s, t1, err = None, None, t0, None
    try:
        s = scene.make_scene(n_cams=poses, n_pts=100 * poses, n_obs=1000 * poses, seed=0x5EED0008)
        t1 = time.perf_counter()
        m = WindowedMap(s, windows, lambda: BundlerLib(False, device=device), lambda b, sc: load_scene(b, sc, bulk=True), rank=rank, world=world,
                        dist=dist, overlap=overlap, exchange_device=D.stats_device(device), threads=threads, device=device)
    except Exception as e:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        err = f"{type(e).__name__}: {e}"
    if dist is not None:
        flag = torch.tensor([0.0 if err is None else 1.0], dtype=torch.float64, device=D.stats_device(device))
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if flag.item() != 0.0 and err is None:
            err = "another rank could not build its windows"
    if err is not None:
        return {"error": err}
    t2 = time.perf_counter()
    return _strong_scaling_run(m, s, dist, device, rank, world, poses, windows, overlap, iters, warmup, threads, t0, t1, t2)


def _strong_scaling_run(m, s, dist, device, rank, world, poses, windows, overlap, iters, warmup, threads, t0, t1, t2):
    import torch
    from mageslam_amd import dist as D
    failure = []

    def outer():
        # a rank whose windows fail to step still takes part in the exchange: the ranks stay in lockstep and ALL report the failure
        try:
            return m.outer_iteration(HUBER)
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            failure.append(f"{type(e).__name__}: {e}")
            m.exchange()
            return float("nan")

    errs = [outer() for _ in range(warmup)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    errs += [outer() for _ in range(iters)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t3
    failed = 1.0 if failure else 0.0
    if dist is not None:
        t = torch.tensor([el, failed], dtype=torch.float64, device=D.stats_device(device))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el, failed = float(t[0].item()), float(t[1].item())
    for b in m.bundlers.values():
        b.close()
    if failed:
        return {"error": failure[0] if failure else "another rank failed to step its windows"}
    return {"metric": "outer iterations/s of ONE windowed map (every window takes one LM iteration, then the pose-block exchange)",
            "workload": f"one map of {poses} poses / {100 * poses} points / {1000 * poses} observations in {windows} keyframe windows, overlap {overlap}",
            "scaling": "strong", "n_gpus": world, "windows_per_rank": len(m.mine), "windows_in_flight_per_rank": min(threads, max(len(m.mine), 1)),
            "value": round(iters / el, 3), "unit": "outer iterations/s", "ms_per_outer_iteration": round(1e3 * el / iters, 3),
            "lm_window_iterations_per_s": round(iters * windows / el, 2), "exchange_bytes_per_iteration": poses * 64,
            "exchange": ("RCCL all-reduce of the pose block in HBM" if D.init.backend == "nccl" else
                         ("gloo all-reduce through the host" if dist is not None else "single rank: the block never leaves HBM, no collective")),
            "mse_of_rank0_windows": [round(float(e), 5) for e in errs], "scene_s": round(t1 - t0, 1), "cut_and_load_s": round(t2 - t1, 1)}


def run_extras(device) -> dict:
    legs = (("small_shapes", extra_small_shapes), ("small_shapes_cpu_baseline", cpu_baseline_small_shapes),
            ("config3", lambda: extra_config3(device)), ("config3_cpu_baseline", cpu_baseline_config3),
            ("config4_end_to_end", lambda: extra_config4_end_to_end(device)), ("cold_start", lambda: extra_cold_start(device)),
            ("skyline_solve", lambda: extra_skyline_solve(device)), ("sustained", lambda: extra_sustained(device)),
            ("concurrent_handles", lambda: extra_concurrent_handles(device)),
            ("config2", lambda: extra_config2(device)), ("config2_cpu_baseline", cpu_baseline_config2))
    out = {}
    for name, fn in legs:
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001 - an extra is a report, never a reason to lose the headline
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        out[name]["wall_s"] = round(time.perf_counter() - t0, 2)
    # the two shapes under the names the review asked for, device path and CPU oracle side by side
    hip, cpu = out.pop("small_shapes", {}), out.pop("small_shapes_cpu_baseline", {})
    for name in SMALL_SHAPES:
        out[name] = {"hip": hip.get(name, hip.get("error")), "cpu_oracle_1_core": cpu.get(name, cpu.get("error"))}
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="global", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra configurations measured after the headline")
    ap.add_argument("--submaps-per-gpu", type=int, default=2,
                    help="replica mode: K independent sub-maps stepped concurrently per GPU, reported as a LABELLED SECONDARY beside the headline (default 2: "
                         "a second sub-map's update-bound launches fill the chain-bound tail of the first one's factorisation; 4 measured lower than 2 -- the "
                         "chain-bound launches hold one workgroup per compute unit and then queue behind each other; 1 = off)")
    args = ap.parse_args()

    import torch
    from mageslam_amd import dist as D
    info = D.rank_info()
    rank, local_rank, world = info.rank, info.local_rank, info.world
    if args.gpus != world and world > 1:
        print(f"--gpus {args.gpus} != WORLD_SIZE {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X: torch.cuda.is_available() is False", file=sys.stderr)
        return 2
    n_dev = torch.cuda.device_count()
    device = local_rank % max(n_dev, 1)             # one GPU per rank on the node; the modulo only matters for the single-GPU plumbing test
    torch.cuda.set_device(device)
    # "nccl" is RCCL on ROCm.  MAGE_DIST_BACKEND=gloo lets the N > 1 plumbing be exercised on a box with fewer GPUs than ranks.
    dist = D.init(os.environ.get("MAGE_DIST_BACKEND", "nccl"), info, device_index=device)

    from mageslam_amd import scene
    from mageslam_amd.bundler import BundlerLib, load_scene

    kw = dict(WORKLOADS[args.workload])
    kw["seed"] = D.submap_seed(kw["seed"], rank)    # independent sub-map per rank (weak scaling, no data-path collective)
    s = scene.make_scene(**kw)
    b = BundlerLib(False, device=device)
    load_scene(b, s, bulk=True)
    if args.workload in LAMBDA_SEED:
        b.SetCurrentLambda(LAMBDA_SEED[args.workload])
    outl: list = []
    trials = 0
    # W untimed steps: the first uploads the problem and builds the graph structure.  The dense solve's task lists for this size are
    # built by a worker thread while the first factorisations go column by column (extra.cold_start measures that); the timed region
    # is the steady state, so the warm-up waits for them behind its first step.
    import ctypes as _C
    from mageslam_amd.bundler import lib as _lib
    _lib().mage_debug_chol_wait_schedule.restype = _C.c_int
    _lib().mage_debug_chol_wait_schedule.argtypes = [_C.c_int, _C.c_int, _C.POINTER(_C.c_double)]
    for w_i in range(args.warmup):
        b.StepBundleAdjustment([HUBER], 1e30, outl)
        if w_i == 0:
            _ms = _C.c_double(0.0)
            _lib().mage_debug_chol_wait_schedule(device, int(b.profile().padded_order), _C.byref(_ms))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # In the timed region only the dominant kernel group is bracketed by HIP events (two records per factorisation, on the solver's
    # stream): `roofline`.  The stage spans of `roofline_hbm` need seven records per LM trial, each a few microseconds of idle
    # stream; they are taken over STAGE_STEPS further steps of the same handle right after the timed region.
    b.enable_profiling(2)
    barrier()
    t0 = time.perf_counter()
    mse = float("nan")
    for _ in range(args.steps):
        mse = b.StepBundleAdjustment([HUBER], 1e30, outl)
        trials += sum(t["trials"] for t in b.trace())
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed, total_steps, worst_rmse = D.reduce_stats(dist, elapsed, args.steps, float(np.sqrt(mse)), device=D.stats_device(device))
    prof = b.profile()
    STAGE_STEPS = 5
    b.enable_profiling(1)
    for _ in range(STAGE_STEPS):
        b.StepBundleAdjustment([HUBER], 1e30, outl)
    prof_stages = b.profile()
    b.close()                                       # the headline's handle: its buffers go back to the cache before the extras

    # ---- after the timed region: the strong-scaling point of this N (every rank takes part), then rank 0's extras at N = 1
    strong = None
    if not args.no_extras and args.workload == "global":
        t_s = time.perf_counter()
        try:
            # (ranks that SHARE a GPU -- the single-GPU plumbing test -- step their windows one at a time: several processes
            # oversubscribing one device with large solves are time-sliced by the hardware scheduler and run into re-tried trials)
            strong = strong_scaling_windowed(device, dist, rank, world, threads=4 if world <= max(n_dev, 1) else 1)
        except Exception as e:  # noqa: BLE001 - a report, never a reason to lose the headline
            import traceback
            traceback.print_exc()
            strong = {"error": f"{type(e).__name__}: {e}"}
        strong["wall_s"] = round(time.perf_counter() - t_s, 2)
    secondary = None
    if args.submaps_per_gpu > 1 and args.workload == "global" and not args.no_extras:
        # replica mode with K sub-maps per GPU: every rank steps K independent sub-maps from K host threads; aggregate over ranks
        try:
            r = extra_concurrent_handles(device, counts=(args.submaps_per_gpu,), steps=max(args.steps, 4))[str(args.submaps_per_gpu)]
            agg, _, _ = D.reduce_stats(dist, r["ms_per_iteration_per_handle"] * 1e-3, 0, 0.0, device=D.stats_device(device))
            secondary = {"submaps_per_gpu": args.submaps_per_gpu, "n_gpus": world,
                         "lm_iterations_per_s_aggregate": round(world * args.submaps_per_gpu / agg, 2),
                         "note": "LABELLED SECONDARY: K independent 1k-pose sub-maps stepped concurrently per GPU (max over ranks of the per-iteration time); the headline `value` stays one sub-map per GPU"}
        except Exception as e:  # noqa: BLE001
            secondary = {"error": f"{type(e).__name__}: {e}"}
    extra = None
    if rank == 0 and world == 1 and not args.no_extras and args.workload == "global":
        extra = run_extras(device)
        if strong is not None:
            extra["config5_windowed_1gpu"] = strong

    if rank == 0:
        value = total_steps / elapsed
        nfac = max(int(prof.n_factorizations), 1)
        fac_ms = prof.factor_ms_total / nfac
        flops = prof.factor_flops_each
        achieved = flops / (fac_ms * 1e-3) / 1e12 if fac_ms > 0 else 0.0
        traffic = None
        try:   # HBM bytes per factorisation from the committed PMC passes (tools/pmc_to_traffic.py); same padded order only
            tj = json.load(open(_latest_profile("chol_traffic.json")))
            if int(tj.get("n_pad", -1)) == int(prof.padded_order):
                traffic = float(tj["bytes_per_factorization"])
        except Exception:
            traffic = None
        line = {
            "metric": "LM iterations/sec + final reproj RMSE, 1k poses / 100k pts / 1M obs synthetic BA",
            "value": value, "unit": "LM iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "final_reproj_rmse_px": worst_rmse,
            "trials_per_iteration": trials / max(args.steps, 1),
            # health of the dense solve's in-launch hand-offs over the headline handle's life (mage_ba_profile): both 0 unless several
            # PROCESSES oversubscribe the GPU (DESIGN.md "forward progress")
            "stall_counters": {"trials_rerun_after_stall": int(prof.trials_rerun_after_stall),
                               "fallback_to_separate_launches": int(prof.fallback_to_separate_launches)},
            "config": {"workload": f"{args.workload}: {kw['n_cams']} poses / {kw['n_pts']} points / {kw['n_obs']} observations, "
                                   f"Huber {HUBER}, poses 0,1 fixed, lambda seed {LAMBDA_SEED.get(args.workload, 'g2o default')}, "
                                   f"one independent sub-map per GPU",
                       "parallelism": f"replica x{world} (independent sub-maps)",
                       "control_plane": (D.init.backend or "none") + (" (RCCL)" if D.init.backend == "nccl" else "")},
            "roofline": {
                "kernel": "dense reduced-camera Cholesky (k_chol_dag: the factorisation + forward substitution as ONE persistent task-graph launch [f64 MFMA], + its state memset + k_bsolve_persist), "
                          "HIP-event span per factorisation on the solver stream",
                "bound": "mfma", "achieved": achieved, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / F64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                "traffic_source": "committed profile (profiles/r<NN>_chol_traffic.json of the latest round: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over "
                                  "tools/_bin/chol_test at the same padded order, (2*FETCH_SIZE + WRITE_SIZE)*1024 per factorisation; regenerate with "
                                  "tools/regen_profiles.sh) -- NOT measured in this run",
                "flops_per_launch": flops, "ms_per_launch": fac_ms, "launches": int(prof.n_factorizations),
                "system_order": int(prof.system_order), "padded_order": int(prof.padded_order),
                "schur_ms_per_launch": prof_stages.schur_ms_total / max(int(prof_stages.schur_launches), 1),
                "measured_over": "the timed region: one HIP-event pair per factorisation on the solver's stream",
            },
        }
        # the HBM-bound stages of the same iterations: linearise (once per iteration), Schur build and back-substitution + trial
        # evaluation (once per trial); algorithmic bytes from the handle (every array a stage reads or writes counted once)
        ps = prof_stages
        n_lin, n_sch, n_upd = max(int(ps.linearize_launches), 1), max(int(ps.schur_launches), 1), max(int(ps.update_launches), 1)
        stages = {
            "linearize": (ps.linearize_ms_total / n_lin, ps.linearize_bytes_each),
            "schur_build": (ps.schur_ms_total / n_sch, ps.schur_bytes_each),
            "backsubst_and_trial_error": (ps.update_ms_total / n_upd, ps.update_bytes_each),
        }
        tot_ms = sum(ms for ms, _ in stages.values())
        tot_b = sum(by for _, by in stages.values())
        line["roofline_hbm"] = {
            "kernels": "k_small_linearize in its large-problem form (residuals + landmark side with 8 lanes per landmark + camera side, one launch) | k_schur_prepare (skyline zero-fill + landmark inverses + the fold of the linearisation's chi2 partials, one launch) + k_schur_stream (resident wavefronts working through per-compute-unit block lists; its diagonal blocks also give the reduced rhs) | k_pose_update + k_backsub (which also evaluates the trial's residuals) + k_reduce_sum; "
                       f"HIP-event spans on the solver stream, per LM iteration with one trial, over {STAGE_STEPS} further steps of the same handle right "
                       "after the timed region (seven event records per trial cost idle stream time: the timed region keeps only the factorisation's pair)",
            "bound": "hbm", "achieved": tot_b / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (tot_b / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if tot_ms > 0 else 0.0,
            "algorithmic_bytes": tot_b, "ms": tot_ms,
            "stages": {k: {"ms": ms, "algorithmic_bytes": by, "GB/s": by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                           "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0} for k, (ms, by) in stages.items()},
            "traffic": None,
        }
        # The same span against SURVEY.md section 8(d)'s COMPULSORY bytes of these stages (what no implementation can avoid: the 20-byte
        # observation records four times, the points twice read and once written, the poses), so that the fraction cannot drift with
        # the handle's own accounting: the handle counts every array a stage touches (compact W records, list entries, landmark blocks).
        sc = WORKLOADS[args.workload]
        surv_b = 4 * 20.0 * sc["n_obs"] + 3 * 24.0 * sc["n_pts"] + 2 * 96.0 * sc["n_cams"]
        line["roofline_hbm"]["survey_8d"] = {"compulsory_bytes": surv_b, "GB/s": surv_b / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0,
                                             "frac": surv_b / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if tot_ms > 0 else 0.0,
                                             "note": "S itself (written once by the Schur build inside its skyline, read and written by the factorisation) is the dense solve's traffic: roofline.traffic"}
        try:   # HBM bytes per LM iteration of the three stages from the committed PMC passes (tools/pmc_stage_traffic.py)
            sj = json.load(open(_latest_profile("ba_stage_traffic.json")))
            if args.workload == "global":
                hb = line["roofline_hbm"]
                hb["traffic"] = float(sj["total_hbm_bytes_per_iteration"])
                for k, v in sj["stages_hbm_bytes_per_iteration"].items():
                    if k in hb["stages"]:
                        hb["stages"][k]["traffic"] = float(v)
                hb["traffic_source"] = ("committed profile (profiles/r<NN>_ba_stage_traffic.json of the latest round: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over "
                                        "bench.py on the 1k-pose map), not measured in this run")
        except Exception:  # noqa: BLE001 - the profile is optional
            pass
        if not args.no_cpu_baseline and world == 1:   # reported baseline: rank 0, N = 1 only
            try:
                line["cpu_baseline"] = cpu_baseline(args.workload)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "LM iterations/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
        if strong is not None:
            line["strong_scaling"] = strong
        if secondary is not None:
            line["replica_submaps_per_gpu"] = secondary
        if extra is not None:
            line["extra"] = extra
        line["lambda_note"] = ("value is measured with SetCurrentLambda(5e6) (single-trial iterations, SURVEY 8d); from g2o's own lambda start the same "
                               "window takes more trials per iteration: see extra.sustained.g2o_default_lambda")
        print(json.dumps(_finite(line)), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
