"""Shared helpers: run a bundler through a golden case and compare with the expected values."""
import os

import numpy as np

from mageslam_amd.scene import Scene, Tethers

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BA_CASES = ["ba_tiny_clean", "ba_tiny_outliers", "ba_tiny_pose_only", "ba_small_fixedcams"]
# Cases with tether edges.  The reference differentiates the distance / rotation tethers numerically with a step of 1e-9
# (g2o BaseMultiEdge), which turns every last-bit difference of the error function into a ~1e-7 relative difference of the
# Jacobian; implementations that do not share their arithmetic bit for bit (numpy matrices vs C quaternions) therefore agree
# to ~1e-7 on the state, still 100x inside the north star's 1e-5.  HIP vs C oracle share the operation order and are held tighter.
BA_TETHER_CASES = ["ba_tiny_tethers", "ba_tethered_blind_camera"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=True)
    s = Scene(n_cams=len(z["cam_t"]), n_pts=len(z["points"]), n_obs=len(z["obs_uv"]), cam_t=z["cam_t"], cam_R=z["cam_R"],
              cam_K=z["cam_K"], cam_fixed=z["cam_fixed"], points=z["points"], obs_uv=z["obs_uv"], obs_cam=z["obs_cam"],
              obs_pt=z["obs_pt"], obs_info=z["obs_info"])
    if "teth_dist_d" in z.files:
        s.tethers = Tethers(**{k[5:]: z[k] for k in z.files if k.startswith("teth_")})
    return s, z


def quat_to_R(q):
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((len(q), 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - z * w); R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w); R[:, 2, 1] = 2 * (y * z + x * w); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def run_case(bundler, load_scene, name, rtol_state=1e-9, rtol_chi=1e-9):
    """Drives `bundler` (BundlerLib call surface) through the case and asserts against the fixture.

    Tolerances: the north star asks for 1e-5 relative on pose/point estimates; implementations that
    follow the same algorithm in float64 agree far better, so the state is held to `rtol_state`
    (default 1e-9), chi2/lambda to 1e-9 relative, the float32 mean-square-error return to 1e-6 relative,
    and the integer outputs (trial counts, result codes, outlier index lists) exactly.
    """
    s, z = load_case(name)
    load_scene(bundler, s)
    trace, mse, outl, n_out = [], [], [], []
    for hw, thr in zip(z["hubers"], z["thrs"]):
        o = []
        mse.append(bundler.StepBundleAdjustment(np.asarray(hw, np.float32), float(thr), o))
        for t in bundler.trace():
            trace.append([t["code"], t["trials"], t["chi_before"], t["chi_after"], t["lam"]])
        outl.extend(o); n_out.append(len(o))
    trace = np.array(trace)
    exp = z["exp_trace"]
    assert trace.shape == exp.shape
    assert np.array_equal(trace[:, :2], exp[:, :2]), "result codes / trial counts differ"
    np.testing.assert_allclose(trace[:, 2:], exp[:, 2:], rtol=rtol_chi)
    assert n_out == list(z["exp_n_out"])
    assert np.array_equal(np.array(outl, np.uint32), z["exp_outliers"]), "outlier index lists differ"
    np.testing.assert_allclose(np.array(mse, np.float32), z["exp_mse"], rtol=1e-6)
    P = bundler.poses_f64()
    np.testing.assert_allclose(quat_to_R(P[:, :4]), z["exp_R"], atol=rtol_state)
    np.testing.assert_allclose(P[:, 4:], z["exp_t"], rtol=rtol_state, atol=rtol_state)
    np.testing.assert_allclose(bundler.points_f64(), z["exp_X"], rtol=rtol_state, atol=rtol_state)
    # float32 getters (BundlerLib.cpp:457-471)
    t0, R0 = bundler.GetPose(s.n_cams - 1)
    np.testing.assert_allclose(t0, z["exp_t"][-1].astype(np.float32), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(R0.reshape(3, 3).T, z["exp_R"][-1].astype(np.float32), atol=1e-6)
    np.testing.assert_allclose(bundler.GetPoint(3), z["exp_X"][3].astype(np.float32), rtol=1e-6, atol=1e-6)
    return bundler


def random_graph_scene(case):
    """Scene number `case` of the differential structure tests: the observation list shuffled, thinned, with duplicate (camera,
    point) pairs, random fixed flags and (every third case) tethers.  Returns (scene, points_fixed, tethered)."""
    from mageslam_amd import scene as _scene
    rng = np.random.default_rng(0xBA5E + case)
    n_cams = int(rng.integers(2, 14)); n_pts = int(rng.integers(6, 80)); K = int(rng.integers(2, min(n_cams, 6) + 1))
    s = _scene.make_scene(n_cams=n_cams, n_pts=n_pts, n_obs=n_pts * K, seed=0x5EED2000 + case, fixed=(), outlier_frac=0.05 * (case % 3))
    idx = rng.permutation(s.n_obs)                                  # BundleAdjust.cpp feeds observations in map order, not by point
    idx = idx[rng.random(s.n_obs) > 0.1]                            # some points lose observations (a few lose all of them)
    dup = rng.choice(idx, size=max(1, len(idx) // 8))               # the same (camera, point) observed twice
    idx = np.concatenate([idx, dup])
    s.obs_uv, s.obs_cam, s.obs_pt, s.obs_info = s.obs_uv[idx], s.obs_cam[idx], s.obs_pt[idx], s.obs_info[idx]
    s.obs_uv = s.obs_uv.copy(); s.obs_uv[len(idx) - len(dup):] += rng.normal(0, 0.5, (len(dup), 2)).astype(np.float32)
    s.n_obs = len(idx)
    fixed = rng.random(n_cams) < 0.3
    fixed[int(rng.integers(0, n_cams))] = True                      # at least one anchor
    s.cam_fixed = fixed
    tethered = case % 3 == 2 and n_cams >= 3
    if tethered:
        s.tethers = _scene.make_tethers(s, n_dist=2, n_rot=1, n_xf=2, seed=0x7E7E0100 + case)
    points_fixed = case % 8 == 7
    return s, points_fixed, tethered
