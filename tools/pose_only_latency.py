"""Latency of the tracking-thread shape (TrackLocalMap::OptimizeCameraPose, TrackLocalMap.cpp:421-501): one free pose,
300 fixed map points, a few LM iterations per frame.    python tools/pose_only_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mageslam_amd import scene
from mageslam_amd.bundler import BundlerLib, load_scene

s = scene.make_scene(n_cams=1, n_pts=300, n_obs=300, seed=31, fixed=(), outlier_frac=0.05)
for _ in range(3):
    b = BundlerLib(True); load_scene(b, s, bulk=True); b.StepBundleAdjustment([4.0] * 4, 20.25, [])
n = 50
t0 = time.perf_counter()
for _ in range(n):
    b = BundlerLib(True); load_scene(b, s, bulk=True)
    b.StepBundleAdjustment([4.0, 4.0, 4.0, 4.0], 20.25, [])
    b.GetPose(0)
t1 = time.perf_counter()
b = BundlerLib(True); load_scene(b, s, bulk=True); b.StepBundleAdjustment([4.0], 20.25, [])
t2 = time.perf_counter()
for _ in range(200):
    b.StepBundleAdjustment([4.0], 1e30, [])
t3 = time.perf_counter()
print(f"create + load + 4 LM iterations + outlier pass + GetPose: {1e3*(t1-t0)/n:.3f} ms per frame; steady LM iteration {1e3*(t3-t2)/200:.3f} ms")
