"""GPU tests (-m gpu) of the window-sharded map on the product path (BASELINE.json configs[4], second form):
* the HIP back-end driven by mageslam_amd/windowed.py (pose block in HBM) against the same driver on the CPU oracle;
* the C++ driver (include/mage_window.h) against the Python driver: same pose block bit for bit, for any thread count;
* tools/windowed_rccl.cpp: the C++ driver with ncclAllReduce on the device block (one rank always; two when the box has two GPUs);
* two ranks sharing the one GPU of the test box under torch.distributed.run (gloo carries the block there);
* the full-size map (8 000 poses / 800 k points / 8 M observations in 8 windows) on one GPU."""
import hashlib
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from mageslam_amd import scene
from mageslam_amd.bundler import BundlerLib, load_scene
from mageslam_amd.windowed import WindowedMap
from oracle.oracle import OracleBundler, load_scene_bulk

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENE = dict(n_cams=48, n_pts=960, n_obs=7680, seed=0x5EED0B10)


def _hip_load(b, s):
    load_scene(b, s, bulk=True)


def test_update_camera_poses_equals_a_rebuilt_bundler():
    """The extension re-seeds poses without touching the graph; the next steps must equal those of a bundler built from
    scratch on the edited scene (same lambda start), on the HIP path and on the oracle."""
    s = scene.make_scene(n_cams=12, n_pts=150, n_obs=900, seed=0x5EED0B20, fixed=(0, 1, 10, 11))
    rng = np.random.default_rng(5)
    idx = np.array([3, 10, 11], np.uint32)
    t_new = s.cam_t[idx] + rng.normal(0, 0.02, (3, 3)).astype(np.float32)
    R_new = s.cam_R_colmajor()[idx]
    for make, load in ((lambda: BundlerLib(False), _hip_load), (lambda: OracleBundler(False), load_scene_bulk)):
        a = make(); load(a, s)
        a.StepBundleAdjustment([1.8], 1e30, [])                 # estimate now lives on the device
        tA, RA = a.GetPosesBulk(); pA = a.points_f64()
        a.UpdateCameraPoses(idx, t_new, R_new)
        s2 = scene.make_scene(n_cams=12, n_pts=150, n_obs=900, seed=0x5EED0B20, fixed=(0, 1, 10, 11))
        s2.cam_t = tA.copy(); s2.cam_R = RA.reshape(-1, 3, 3).transpose(0, 2, 1).copy(); s2.points = pA.astype(np.float32)
        s2.cam_t[idx] = t_new; s2.cam_R[idx] = s.cam_R[idx]
        b = make(); load(b, s2)
        # b starts from float32 copies of a's float64 estimate (1e-6 m on a 10 m scene = 1e-4 px): agreement is at that level
        for _ in range(2):
            ma = a.StepBundleAdjustment([1.8], 1e30, []); mb = b.StepBundleAdjustment([1.8], 1e30, [])
            assert a.trace()[0]["trials"] == b.trace()[0]["trials"]
        assert abs(ma - mb) < 1e-3 * mb
        np.testing.assert_allclose(a.poses_f64(), b.poses_f64(), atol=1e-4)
        ta, _ = a.GetPosesBulk()
        np.testing.assert_allclose(ta[[10, 11]], t_new[1:], atol=1e-7)          # fixed cameras keep the re-seeded pose


def _run(backend, n_windows, iters, rank=0, world=1, dist=None, threads=1):
    s = scene.make_scene(**SCENE)
    make, load = ((lambda: BundlerLib(False, device=0)), _hip_load) if backend == "hip" else ((lambda: OracleBundler(False)), load_scene_bulk)
    m = WindowedMap(s, n_windows, make, load, rank=rank, world=world, dist=dist, overlap=2, threads=threads,
                    device=0 if backend == "hip" else None)
    mse = [m.outer_iteration(1.8) for _ in range(iters)]
    return m, mse


def _sha(block):
    return hashlib.sha256(np.ascontiguousarray(block, np.float64).tobytes()).hexdigest()


def test_hip_windows_match_the_same_driver_on_the_oracle():
    """Same driver, same float64 pose block, HIP vs CPU oracle underneath: eight outer iterations of four windows agree to
    1e-9 (the single-window tolerance; the block never passes through float32)."""
    mh, eh = _run("hip", 4, 8)
    mo, eo = _run("oracle", 4, 8)
    assert mh.on_device and not mo.on_device
    np.testing.assert_allclose(eh, eo, rtol=1e-6)                     # float32 returns of StepBundleAdjustment
    np.testing.assert_allclose(mh.pose_block(), mo.pose_block(), rtol=1e-9, atol=1e-9)
    assert eh[-1] < eh[1]


@pytest.mark.parametrize("threads", [1, 3])
def test_cpp_driver_equals_python_driver_bit_for_bit(threads):
    """include/mage_window.h (host C++, exchange in HBM) and mageslam_amd/windowed.py give the same pose block, bit for bit,
    and the windows may be stepped from any number of threads."""
    from mageslam_amd.wmap import WindowMap
    mp, ep = _run("hip", 4, 6)
    s = scene.make_scene(**SCENE)
    mc = WindowMap(s, 4, overlap=2, device=0, threads=threads)
    for w, win in enumerate(mp.windows):
        i = mc.window_info(w)
        assert (i["own"], i["cams"], i["pts"], i["obs"], i["mine"]) == (len(win.own), len(win.cams), len(win.pts), len(win.obs), True)
    ec = [mc.outer_iteration(1.8) for _ in range(6)]
    assert ec == ep
    assert _sha(mc.pose_block()) == _sha(mp.pose_block())
    # a window's own state after the run: identical too (own keyframes are never written by the exchange)
    k = len(mp.windows[1].own)
    assert np.array_equal(mc.window_bundler(1).poses_f64()[:k], mp.bundlers[1].poses_f64()[:k])


def _tool(name):
    exe = os.path.join(ROOT, "tools", "_bin", name)
    if not os.path.exists(exe):
        import __graft_entry__ as G
        G.build_tools()
    if not os.path.exists(exe):
        pytest.skip(f"tools/_bin/{name} was not built (no RCCL on the build machine)")
    return exe


def _gpu_count():
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    n = ctypes.c_int(0)
    return n.value if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 else 0


def _run_rccl(tmp_path, world, scene_path, n_windows, overlap, iters, threads, tag):
    exe = _tool("windowed_rccl")
    idf, outp = str(tmp_path / f"id_{tag}"), str(tmp_path / f"block_{tag}")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if world == 1:
            env["MAGE_ALLOW_SINGLE_RANK"] = "1"          # a one-rank communicator on purpose (the tool refuses an accidental one)
        procs.append(subprocess.Popen([exe, scene_path, str(n_windows), str(overlap), str(iters), "1.8", str(threads), idf, outp],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=900)
        assert p.returncode == 0, (o + e)[-3000:]
        outs.append(o)
    info = json.loads([l for l in outs[0].splitlines() if l.startswith("{")][-1])
    blocks = [np.fromfile(f"{outp}.rank{r}.bin", np.float64).reshape(-1, 8) for r in range(world)]
    return info, blocks


def test_rccl_driver_equals_python_driver(tmp_path):
    """tools/windowed_rccl.cpp: the C++ driver with the pose block all-reduced by ncclAllReduce in HBM.  One rank exercises
    the RCCL path on any box; with two GPUs two ranks (one per GPU) must give the same block on both ranks, bit for bit."""
    mp, ep = _run("hip", 4, 6)
    want = _sha(mp.pose_block())
    path = str(tmp_path / "scene.bin")
    scene.save_scene(scene.make_scene(**SCENE), path)
    info, blocks = _run_rccl(tmp_path, 1, path, 4, 2, 6, 2, "w1")
    assert info["allreduce_calls"] == 6 and info["mse_rank0"] == pytest.approx(ep, rel=1e-7) and info["comm_nranks"] == 1
    assert _sha(blocks[0]) == want
    if _gpu_count() >= 2:
        info, blocks = _run_rccl(tmp_path, 2, path, 4, 2, 6, 1, "w2")
        assert info["comm_nranks"] == 2
        assert _sha(blocks[0]) == want and _sha(blocks[1]) == want


WORKER = textwrap.dedent("""
    import sys, json, hashlib
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    from mageslam_amd import dist as D
    from test_windowed_gpu import _run, _sha
    info = D.rank_info()
    dist = D.init("gloo", info)
    m, mse = _run("hip", 4, 5, rank=info.rank, world=info.world, dist=dist)
    import os
    with open(os.path.join(os.environ["MAGE_TEST_OUT"], "rank%%d.json" %% info.rank), "w") as f:      # not stdout: two ranks' lines can interleave
        json.dump(dict(rank=info.rank, sha=_sha(m.pose_block()), mine=m.mine, on_device=m.on_device), f)
    dist.barrier(); dist.destroy_process_group()
""") % (ROOT, os.path.join(ROOT, "tests"))


def test_two_ranks_sharing_the_gpu_equal_one_rank_bit_for_bit(tmp_path):
    m, _ = _run("hip", 4, 5)
    want = _sha(m.pose_block())
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    for attempt in range(2):            # the port is free when probed, not reserved: one retry if something else took it meanwhile
        sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, MAGE_TEST_OUT=str(tmp_path)))
        if p.returncode == 0:
            break
    assert p.returncode == 0, p.stderr[-3000:]
    outs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert [o["mine"] for o in outs] == [[0, 1], [2, 3]] and all(o["on_device"] for o in outs)
    assert outs[0]["sha"] == want and outs[1]["sha"] == want


def test_full_size_map_in_eight_windows():
    """BASELINE.json configs[4], second form, at its real size on one GPU: 8 000 poses / 800 k points / 8 M observations cut
    into 8 windows (overlap 10), driven by the C++ driver.
    * the pose block is bit-identical whether the windows are stepped by 1 thread or by 4 concurrent ones;
    * the error does not increase from the first exchange on;
    * a window's first LM iteration equals a stand-alone BundlerLib on that window's problem, bit for bit."""
    from mageslam_amd.wmap import WindowMap
    from mageslam_amd.windowed import cut_windows
    s = scene.make_scene(n_cams=8000, n_pts=800_000, n_obs=8_000_000, seed=0x5EED0008)
    m1 = WindowMap(s, 8, overlap=10, device=0, threads=1)
    info = m1.window_info(3)
    assert info["own"] == 1000 and info["cams"] > 1020 and info["obs"] > 1_000_000
    e1 = [m1.outer_iteration(1.8)]
    # stand-alone bundler on window 3 (cut by the Python twin of the C++ cutter)
    win = cut_windows(s, 8, overlap=10)[3]
    assert (len(win.own), len(win.cams), len(win.pts), len(win.obs)) == (info["own"], info["cams"], info["pts"], info["obs"])
    alone = BundlerLib(False, device=0)
    _hip_load(alone, win.scene)
    mse_alone = alone.StepBundleAdjustment([1.8], 1e30, [])
    wb = m1.window_bundler(3)
    k = len(win.own)
    assert np.array_equal(wb.poses_f64()[:k], alone.poses_f64()[:k]) and np.array_equal(wb.points_f64(), alone.points_f64())
    assert wb.trace()[0]["chi_after"] == alone.trace()[0]["chi_after"] and np.isfinite(mse_alone)
    alone.close(); del win
    e1 += [m1.outer_iteration(1.8) for _ in range(4)]
    b1 = m1.pose_block()
    m1.close()
    m4 = WindowMap(s, 8, overlap=10, device=0, threads=4)
    e4 = [m4.outer_iteration(1.8) for _ in range(5)]
    assert e4 == e1 and _sha(m4.pose_block()) == _sha(b1)
    assert all(b <= a * (1 + 1e-6) for a, b in zip(e1[1:], e1[2:])) and e1[-1] < e1[0]
    assert np.all(np.abs(np.linalg.norm(b1[:, :4], axis=1) - 1) < 1e-12)            # every row was published by exactly one window
    m4.close()
