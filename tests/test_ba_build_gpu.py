"""GPU tests (-m gpu) of the graph-structure build (mageslam_amd/csrc/ba_build.hip).

What is built is g2o's initializeOptimization + buildStructure (index maps, active edges, Hessian blocks; BundlerLib.cpp:156-166,
SURVEY appendix A.5) as flat lists.  The device build and its host twin (MAGE_BA_BUILD=host) must produce the SAME lists,
element for element -- orders come from keys, not from arrival -- and therefore bit-identical solves."""
import os

import numpy as np
import pytest

from mageslam_amd import scene
from mageslam_amd.bundler import BundlerLib, load_scene

from ba_cases import random_graph_scene

pytestmark = pytest.mark.gpu


class _build_mode:
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.old = os.environ.get("MAGE_BA_BUILD")
        os.environ["MAGE_BA_BUILD"] = self.mode

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("MAGE_BA_BUILD", None)
        else:
            os.environ["MAGE_BA_BUILD"] = self.old


def _run(s, points_fixed, mode, calls):
    with _build_mode(mode):
        b = BundlerLib(points_fixed)
        load_scene(b, s, bulk=True)
        out, mses = [], []
        mses.append(b.StepBundleAdjustment(np.asarray(calls[0][0], np.float32), calls[0][1], out))
        lists = {n: b.structure(n).copy() for n in BundlerLib.STRUCTURE_LISTS}
        sizes = b.structure("sizes").copy()
        for hw, thr in calls[1:]:
            mses.append(b.StepBundleAdjustment(np.asarray(hw, np.float32), thr, out))
        return lists, sizes, out, mses, b.poses_f64(), b.points_f64()


def _assert_same_build(s, points_fixed, calls):
    lh, zh, oh, mh, Ph, Xh = _run(s, points_fixed, "host", calls)
    ld, zd, od, md, Pd, Xd = _run(s, points_fixed, "device", calls)
    assert zh[9] == 0 and zd[9] == 1, "the build mode was not honoured"
    assert np.array_equal(zh[:9], zd[:9]), (zh, zd)
    for n in BundlerLib.STRUCTURE_LISTS:
        a, b = lh[n], ld[n]
        assert a.shape == b.shape, (n, a.shape, b.shape)
        if not np.array_equal(a, b):
            bad = np.nonzero(a != b)[0]
            raise AssertionError(f"list {n}: {len(bad)} of {a.size} entries differ, first at {bad[0]}: host {a[bad[0]]} device {b[bad[0]]}")
    # identical lists, identical arithmetic: the solves are bitwise equal
    assert oh == od
    assert np.array_equal(np.array(mh), np.array(md), equal_nan=True)
    assert np.array_equal(Ph, Pd) and np.array_equal(Xh, Xd)
    return zd


@pytest.mark.parametrize("case", range(24))
def test_device_build_equals_host_build_on_random_graphs(case):
    s, points_fixed, _ = random_graph_scene(case)
    _assert_same_build(s, points_fixed, [([1.8], 30.0), ([0.9, 0.9], 16.0), ([0.9], 9.0)])


def test_device_build_on_the_golden_tiny_and_empty_edges():
    s = scene.make_config("tiny", outlier_frac=0.02)
    _assert_same_build(s, False, [([1.8], 7.25), ([1.8], 6.5)])
    _assert_same_build(s, True, [([1.8], 7.25), ([1.8], 6.5)])
    # a camera and a point without any observation, and a point seen only by fixed cameras
    keep = (s.obs_cam != 3) & (s.obs_pt != 17)
    s.obs_uv, s.obs_cam, s.obs_pt, s.obs_info, s.n_obs = s.obs_uv[keep], s.obs_cam[keep], s.obs_pt[keep], s.obs_info[keep], int(keep.sum())
    _assert_same_build(s, False, [([1.8], 1e30), ([1.8], 1e30)])


def test_device_build_local_ba_config():
    """BASELINE.json configs[2]: 20 keyframes / 5k points / 50k observations, 7 fixed keyframes -- 13 rows of S with ~2 700 slots
    each, the shape where a row's contributions are split over eight wavefronts."""
    s = scene.make_config("local", outlier_frac=0.02)
    z = _assert_same_build(s, False, [([0.9], 7.25), ([0.9], 6.5), ([0.9], 5.9)])
    assert z[2] == 13 and z[0] == 50000


def test_device_build_long_tracks_and_many_cameras():
    """Points seen by up to 40 cameras (the rank inside a landmark's bucket walks 40 keys, a lane of the row split holds up to 40
    columns) on 150 free cameras (900 x 900 reduced system: 8 tiles, multi-workgroup scans)."""
    s = scene.make_scene(n_cams=152, n_pts=2000, n_obs=2000 * 40, seed=0x5EED3001, spacing=0.02)
    rng = np.random.default_rng(5)
    idx = rng.permutation(s.n_obs)
    s.obs_uv, s.obs_cam, s.obs_pt, s.obs_info = s.obs_uv[idx], s.obs_cam[idx], s.obs_pt[idx], s.obs_info[idx]
    z = _assert_same_build(s, False, [([1.8], 1e30), ([1.8], 1e30)])
    assert z[2] == 150


def test_device_build_more_than_1024_free_cameras():
    """1 300 cameras (1 298 free): the camera scan, the split offsets and the row scans run more than one round of their 1 024-thread
    workgroups, a row of S has up to 1 298 columns (one wavefront per row in the fill kernel)."""
    s = scene.make_scene(n_cams=1300, n_pts=13000, n_obs=13000 * 8, seed=0x5EED3002, spacing=0.1)
    z = _assert_same_build(s, False, [([1.8], 1e30), ([1.8], 1e30)])
    assert z[2] == 1298 and z[0] == 104000


def test_device_build_global_config():
    """BASELINE.json configs[3]: 1k poses / 100k points / 1M observations, 5.5 M Schur contributions in ~21 k blocks."""
    s = scene.make_config("global")
    z = _assert_same_build(s, False, [([1.8], 1e30), ([1.8], 1e30)])
    assert z[0] == 1000000 and z[2] == 998 and z[1] == 100000


def test_default_build_is_on_the_device_for_free_points():
    s = scene.make_config("local")
    os.environ.pop("MAGE_BA_BUILD", None)
    b = BundlerLib(False)
    load_scene(b, s, bulk=True)
    b.StepBundleAdjustment([0.9], 1e30, [])
    assert b.structure("sizes")[9] == 1
