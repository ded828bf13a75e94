"""GPU tests (-m gpu) of the landmark-sharded solve of one map (SURVEY.md 8e "exact algorithm"; include/mage_ba.h:
mage_ba_set_landmark_shard).  The ranks run as threads of this process, several handles on the one GPU, their all-reduce is
mage_device_allreduce_local -- the same code path a multi-GPU run takes except for who adds the buffers.  What must hold: every
rank ends with bit-identical cameras; the map equals the single-handle solve's and the CPU oracle's (integer outputs exactly,
float64 state to 1e-9: the sums are taken in another order); one rank alone is bit-identical to the unsharded handle."""
import ctypes as C

import numpy as np
import pytest

from mageslam_amd import scene, sharded
from mageslam_amd._lib import MAGE_OK, lib
from mageslam_amd.bundler import BundlerLib, load_scene
from oracle.oracle import OracleBundler, load_scene_bulk

pytestmark = pytest.mark.gpu


def _bulk(b, s):
    load_scene(b, s, bulk=True)


def _single(s, calls, points_fixed=False):
    g = BundlerLib(points_fixed)
    _bulk(g, s)
    outs, mses, traces = [], [], []
    for hubers, thr in calls:
        o = []
        mses.append(g.StepBundleAdjustment(hubers, thr, o)); outs.append(o); traces.append(g.trace())
    return dict(poses=g.poses_f64(), points=g.points_f64(), outliers=outs, mse=mses, traces=traces, lam=g.GetCurrentLambda())


def _check_against(res, ref, n_ranks, rtol):
    for r in range(1, n_ranks):
        assert np.array_equal(res["poses"][r], res["poses"][0]), "the ranks' cameras differ"
        assert res["lambdas"][r] == res["lambdas"][0]
        assert res["mse"][r] == res["mse"][0] or all(np.isnan(a) and np.isnan(b) for a, b in zip(res["mse"][r], res["mse"][0]))
        assert [[(t["code"], t["trials"], t["chi_after"]) for t in c] for c in res["traces"][r]] == \
               [[(t["code"], t["trials"], t["chi_after"]) for t in c] for c in res["traces"][0]]
    assert res["outliers"] == ref["outliers"], "outlier lists differ"
    for a, b in zip(res["mse"][0], ref["mse"]):
        assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-6 * abs(b) + 1e-12
    for ca, cb in zip(res["traces"][0], ref["traces"]):
        assert [(t["code"], t["trials"]) for t in ca] == [(t["code"], t["trials"]) for t in cb]
        for a, b in zip(ca, cb):
            assert abs(a["chi_after"] - b["chi_after"]) <= rtol * b["chi_after"] + 1e-12
            assert abs(a["chi_before"] - b["chi_before"]) <= rtol * b["chi_before"] + 1e-12
            assert abs(a["lam"] - b["lam"]) <= rtol * b["lam"]
    np.testing.assert_allclose(res["poses"][0], ref["poses"], rtol=rtol, atol=rtol)
    np.testing.assert_allclose(res["points"], ref["points"], rtol=rtol, atol=rtol)


CALLS = [([1.8], 25.0), ([0.9, 0.9], 16.0), ([0.9], 9.0)]


def test_one_rank_is_the_unsharded_solve_bit_for_bit():
    s = scene.make_scene(n_cams=45, n_pts=4500, n_obs=45000, seed=0x5EED0C00, outlier_frac=0.01)
    res = sharded.solve_on_threads(s, 1, BundlerLib, _bulk, CALLS)
    ref = _single(s, CALLS)
    assert np.array_equal(res["poses"][0], ref["poses"]) and np.array_equal(res["points"], ref["points"])
    assert res["outliers"] == ref["outliers"] and res["lambdas"][0] == ref["lam"]
    assert res["group"].calls > 0


@pytest.mark.parametrize("n_ranks", [2, 3, 5])
def test_ranks_agree_and_match_the_single_handle_solve(n_ranks):
    """45 free cameras (270 -> 3 tiles of the factorisation), 1 % outliers removed between the steps: the re-initialisation after
    a removal must happen on every rank, whichever rank's observation it was."""
    s = scene.make_scene(n_cams=45, n_pts=4500, n_obs=45000, seed=0x5EED0C00 + n_ranks, outlier_frac=0.01)
    res = sharded.solve_on_threads(s, n_ranks, BundlerLib, _bulk, CALLS)
    _check_against(res, _single(s, CALLS), n_ranks, 1e-9)
    assert sum(len(o) for o in res["outliers"]) > 0
    # the work is balanced by k (k + 1) / 2
    w = sharded.landmark_weights(s.obs_pt, s.n_pts)
    loads = np.bincount(res["owner"], weights=w, minlength=n_ranks)
    assert loads.max() - loads.min() <= w.max()


def test_sharded_map_matches_the_oracle():
    s = scene.make_scene(n_cams=30, n_pts=1500, n_obs=12000, seed=0x5EED0C10, outlier_frac=0.02)
    res = sharded.solve_on_threads(s, 3, BundlerLib, _bulk, CALLS)
    o = OracleBundler(False)
    load_scene_bulk(o, s)
    outs, mses, traces = [], [], []
    for hubers, thr in CALLS:
        oo = []
        mses.append(o.StepBundleAdjustment(hubers, thr, oo)); outs.append(oo); traces.append(o.trace())
    _check_against(res, dict(poses=o.poses_f64(), points=o.points_f64(), outliers=outs, mse=mses, traces=traces), 3, 1e-9)


def test_tethers_fixed_cameras_and_a_camera_nobody_on_a_rank_sees():
    """Tether edges are dealt out to the ranks like landmarks (every term of the system is additive); a fixed camera is fixed on
    every rank; with 4 ranks and short tracks most ranks see only some of the cameras -- all of them stay in every rank's system."""
    s = scene.make_scene(n_cams=36, n_pts=300, n_obs=1500, seed=0x5EED0C20, fixed=(0, 7), outlier_frac=0.0)
    s.tethers = scene.make_tethers(s, n_dist=3, n_rot=2, n_xf=3, seed=0x7E7E0C20)
    calls = [([1.8], 1e30), ([0.9], 1e30)]
    res = sharded.solve_on_threads(s, 4, BundlerLib, _bulk, calls)
    _check_against(res, _single(s, calls), 4, 1e-6)        # (the numerically differentiated tethers: as in test_ba_gpu)


def test_rejected_trials_are_rejected_on_every_rank():
    """A tiny user lambda on a badly perturbed scene forces rejected trials: the decision is taken from all-reduced scalars."""
    s = scene.make_scene(n_cams=24, n_pts=480, n_obs=4800, seed=0x5EED0C30, cam_sigma=0.6, rot_sigma=0.15, pt_sigma=1.0)
    calls = [([1.8], 1e30)] * 5
    res = sharded.solve_on_threads(s, 2, BundlerLib, _bulk, calls, prepare=lambda b: b.SetCurrentLambda(1e-9))
    g = BundlerLib(False)
    _bulk(g, s)
    g.SetCurrentLambda(1e-9)
    outs, mses, traces = [], [], []
    for hubers, thr in calls:
        o = []
        mses.append(g.StepBundleAdjustment(hubers, thr, o)); outs.append(o); traces.append(g.trace())
    assert max(t["trials"] for c in traces for t in c) > 1
    _check_against(res, dict(poses=g.poses_f64(), points=g.points_f64(), outliers=outs, mse=mses, traces=traces), 2, 1e-7)


@pytest.mark.parametrize("case", range(16))
def test_randomised_graph_shapes(case):
    """The shapes of test_ba_gpu.py's differential test -- shuffled and thinned observation lists, duplicate (camera, point) pairs,
    random fixed flags, points that lose all observations, cameras nobody observes, tethers, fixed points -- split over 2..4 ranks
    (more ranks than some shapes have landmarks per rank) against the single handle, which takes its small-problem path here."""
    rng = np.random.default_rng(0x5A4D + case)
    n_cams = int(rng.integers(3, 14)); n_pts = int(rng.integers(6, 80)); K = int(rng.integers(2, min(n_cams, 6) + 1))
    s = scene.make_scene(n_cams=n_cams, n_pts=n_pts, n_obs=n_pts * K, seed=0x5EED3000 + case, fixed=(), outlier_frac=0.05 * (case % 3))
    idx = rng.permutation(s.n_obs)
    idx = idx[rng.random(s.n_obs) > 0.1]
    dup = rng.choice(idx, size=max(1, len(idx) // 8))
    idx = np.concatenate([idx, dup])
    s.obs_uv, s.obs_cam, s.obs_pt, s.obs_info = s.obs_uv[idx], s.obs_cam[idx], s.obs_pt[idx], s.obs_info[idx]
    s.obs_uv = s.obs_uv.copy(); s.obs_uv[len(idx) - len(dup):] += rng.normal(0, 0.5, (len(dup), 2)).astype(np.float32)
    s.n_obs = len(idx)
    fixed = rng.random(n_cams) < 0.3
    fixed[int(rng.integers(0, n_cams))] = True
    if fixed.all():
        fixed[int(rng.integers(0, n_cams))] = False                 # a sharded map needs a free camera
    s.cam_fixed = fixed
    tethered = case % 3 == 2
    if tethered:
        s.tethers = scene.make_tethers(s, n_dist=2, n_rot=1, n_xf=2, seed=0x7E7E0200 + case)
    points_fixed = case % 8 == 7
    n_ranks = 2 + case % 3
    calls = [([1.8], 30.0), ([0.9, 0.9], 16.0), ([0.9], 9.0)]
    res = sharded.solve_on_threads(s, n_ranks, lambda: BundlerLib(points_fixed), _bulk, calls)
    _check_against(res, _single(s, calls, points_fixed), n_ranks, 1e-6 if tethered else 1e-8)


def test_full_size_map_on_two_ranks():
    """BASELINE.json configs[3] (1 000 poses / 100 k points / 1 M observations): 148 MB per exchange."""
    s = scene.make_config("global")
    calls = [([0.9], 1e30)]
    res = sharded.solve_on_threads(s, 2, BundlerLib, _bulk, calls)
    _check_against(res, _single(s, calls), 2, 1e-9)
    n_pad = ((6 * int((~s.cam_fixed).sum()) + 127) // 128) * 128
    trials = sum(t["trials"] for t in res["traces"][0][0])
    assert res["group"].doubles >= trials * (n_pad * (n_pad + 128) // 2 + n_pad)


def test_argument_errors():
    L = lib()
    L.mage_ba_set_landmark_shard.argtypes = [C.c_void_p, C.c_int, C.c_int, sharded.ALLREDUCE_FN, C.c_void_p]
    g = BundlerLib(False)
    cb = sharded.ALLREDUCE_FN(lambda *a: 0)
    null = C.cast(None, sharded.ALLREDUCE_FN)
    assert L.mage_ba_set_landmark_shard(g._h, 2, 2, cb, None) != MAGE_OK
    assert L.mage_ba_set_landmark_shard(g._h, 0, 2, null, None) != MAGE_OK
    assert L.mage_ba_set_landmark_shard(g._h, 0, 0, null, None) == MAGE_OK          # off
    # no free camera: refused (the landmarks are independent)
    s = scene.make_scene(n_cams=6, n_pts=50, n_obs=200, seed=3, fixed=tuple(range(6)))
    assert L.mage_ba_set_landmark_shard(g._h, 0, 1, cb, None) == MAGE_OK
    _bulk(g, s)
    with pytest.raises(Exception, match="free camera"):
        g.StepBundleAdjustment([1.0], 1e30, [])


# ---------------------------------------------------------------------------------------------------------------------------
# one process per rank
# ---------------------------------------------------------------------------------------------------------------------------
import hashlib   # noqa: E402
import json      # noqa: E402
import os        # noqa: E402
import socket    # noqa: E402
import subprocess  # noqa: E402
import sys       # noqa: E402
import textwrap  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROC_SCENE = dict(n_cams=45, n_pts=4500, n_obs=45000, seed=0x5EED0C40, outlier_frac=0.01)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _gpu_count():
    hip = C.CDLL("libamdhip64.so")
    n = C.c_int(0)
    return n.value if hip.hipGetDeviceCount(C.byref(n)) == 0 else 0


def _run_rccl(tmp_path, world, scene_path, steps, huber, thr, tag):
    exe = os.path.join(ROOT, "tools", "_bin", "sharded_rccl")
    if not os.path.exists(exe):
        import __graft_entry__ as G
        G.build_tools()
    if not os.path.exists(exe):
        pytest.skip("tools/_bin/sharded_rccl was not built (no RCCL on the build machine)")
    idf, outp = str(tmp_path / f"id_{tag}"), str(tmp_path / f"state_{tag}")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if world == 1:
            env["MAGE_ALLOW_SINGLE_RANK"] = "1"          # a one-rank communicator on purpose (the tool refuses an accidental one)
        procs.append(subprocess.Popen([exe, scene_path, str(steps), str(huber), str(thr), idf, outp], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=900)
        assert p.returncode == 0, (o + e)[-3000:]
        outs.append(o)
    info = json.loads([l for l in outs[0].splitlines() if l.startswith("{")][-1])
    n_cams = PROC_SCENE["n_cams"]
    poses, points = [], np.zeros((PROC_SCENE["n_pts"], 3))
    for r in range(world):
        raw = np.fromfile(f"{outp}.rank{r}.bin", np.float64)
        poses.append(raw[: n_cams * 7].reshape(n_cams, 7))
        rows = raw[n_cams * 7:].reshape(-1, 4)
        points[rows[:, 0].astype(np.int64)] = rows[:, 1:]
    return info, poses, points


def test_cpp_driver_with_rccl(tmp_path):
    """tools/sharded_rccl.cpp: ncclAllReduce on the packed system, on the solver's stream.  One rank runs on any box and must
    equal the unsharded solve bit for bit; with two GPUs, two ranks must agree with each other and with the threads form."""
    s = scene.make_scene(**PROC_SCENE)
    path = str(tmp_path / "scene.bin")
    scene.save_scene(s, path)
    calls = [([0.9], 16.0)] * 3
    ref = _single(s, calls)
    info, poses, points = _run_rccl(tmp_path, 1, path, 3, 0.9, 16.0, "w1")
    assert info["comm_nranks"] == 1 and info["world"] == 1
    assert np.array_equal(poses[0], ref["poses"]) and np.array_equal(points, ref["points"])
    assert info["own_outliers"] == sum(len(o) for o in ref["outliers"]) and info["allreduce_calls"] > 0
    assert info["mse"] == pytest.approx(ref["mse"], rel=1e-6)
    if _gpu_count() >= 2:
        info, poses, points = _run_rccl(tmp_path, 2, path, 3, 0.9, 16.0, "w2")
        assert info["comm_nranks"] == 2
        res = sharded.solve_on_threads(s, 2, BundlerLib, _bulk, calls)
        assert np.array_equal(poses[0], poses[1])
        assert np.array_equal(poses[0], res["poses"][0]) and np.array_equal(points, res["points"])


def test_cpp_driver_refuses_an_accidental_single_rank(tmp_path):
    """A run meant to be collective that comes up with ONE rank (no WORLD_SIZE in the environment) must not pass vacuously."""
    exe = os.path.join(ROOT, "tools", "_bin", "sharded_rccl")
    if not os.path.exists(exe):
        pytest.skip("tools/_bin/sharded_rccl was not built (no RCCL on the build machine)")
    path = str(tmp_path / "scene.bin")
    scene.save_scene(scene.make_scene(**PROC_SCENE), path)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MAGE_ALLOW_SINGLE_RANK")}
    p = subprocess.run([exe, path, "1", "0.9", "16.0", str(tmp_path / "id_x"), str(tmp_path / "state_x")], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 3 and "ONE rank" in p.stderr, (p.returncode, p.stderr[-500:])


WORKER = textwrap.dedent("""
    import sys, json, os
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    import numpy as np
    from mageslam_amd import dist as D, scene, sharded
    from mageslam_amd.bundler import BundlerLib, load_scene
    from test_sharded_gpu import PROC_SCENE, _sha
    info = D.rank_info()
    dist = D.init("gloo", info)
    s = scene.make_scene(**PROC_SCENE)
    group = sharded.TorchGroup(dist, 0)                      # both ranks on GPU 0: the device buffer is staged through the host for gloo
    sb = sharded.ShardedBundler(s, info.rank, info.world, lambda: BundlerLib(False, 0), lambda b, sc: load_scene(b, sc, bulk=True), group.callback())
    outs, mses = [], []
    for _ in range(3):
        o = []
        mses.append(sb.StepBundleAdjustment([0.9], 16.0, o)); outs.append(o)
    pts = np.zeros((s.n_pts, 3)); sb.points_into(pts)
    np.save(os.path.join(os.environ["MAGE_TEST_OUT"], "points%%d.npy" %% info.rank), pts)
    with open(os.path.join(os.environ["MAGE_TEST_OUT"], "rank%%d.json" %% info.rank), "w") as f:
        json.dump(dict(rank=info.rank, sha=_sha(sb.poses_f64()), outliers=outs, mse=mses, calls=group.calls), f)
    dist.barrier(); dist.destroy_process_group()
""") % (ROOT, os.path.join(ROOT, "tests"))


def test_two_processes_sharing_the_gpu_equal_the_threads_form(tmp_path):
    """The torch.distributed callback on real device buffers (torch.as_tensor over the raw pointer, ExternalStream ordering): two
    processes, gloo, one GPU.  A two-term sum has one order: the result is the threads form's bit for bit."""
    s = scene.make_scene(**PROC_SCENE)
    calls = [([0.9], 16.0)] * 3
    res = sharded.solve_on_threads(s, 2, BundlerLib, _bulk, calls)
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    for attempt in range(2):            # the port is free when probed, not reserved: one retry if something else took it meanwhile
        sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), str(script)], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, MAGE_TEST_OUT=str(tmp_path)))
        if p.returncode == 0:
            break
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    outs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert outs[0]["sha"] == outs[1]["sha"] == _sha(res["poses"][0])
    assert [sorted(a + b) for a, b in zip(outs[0]["outliers"], outs[1]["outliers"])] == res["outliers"]
    assert outs[0]["mse"] == outs[1]["mse"] == res["mse"][0]
    points = np.zeros((s.n_pts, 3))
    for r in range(2):
        pr = np.load(tmp_path / f"points{r}.npy")
        own = res["owner"] == r
        points[own] = pr[own]
    assert np.array_equal(points, res["points"])


NCCL_WORKER = textwrap.dedent("""
    import sys, json, os
    sys.path.insert(0, %r); sys.path.insert(0, %r)
    import numpy as np, torch, torch.distributed as dist
    from mageslam_amd import scene, sharded
    from mageslam_amd.bundler import BundlerLib, load_scene
    from test_sharded_gpu import PROC_SCENE, _sha
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%%s" %% sys.argv[1], rank=0, world_size=1, device_id=torch.device("cuda", 0))
    s = scene.make_scene(**PROC_SCENE)
    group = sharded.TorchGroup(dist, 0)
    sb = sharded.ShardedBundler(s, 0, 1, lambda: BundlerLib(False, 0), lambda b, sc: load_scene(b, sc, bulk=True), group.callback())
    outs = []
    for _ in range(3):
        o = []
        sb.StepBundleAdjustment([0.9], 16.0, o); outs.append(o)
    print("RESULT " + json.dumps(dict(sha=_sha(sb.poses_f64()), outliers=outs, calls=group.calls)), flush=True)
    dist.destroy_process_group()
""") % (ROOT, os.path.join(ROOT, "tests"))


def test_torch_rccl_callback_on_one_rank(tmp_path):
    """The RCCL form of the torch.distributed callback (dist.all_reduce on a tensor aliasing the solver's buffer, issued under the
    solver's stream as an ExternalStream, no host synchronisation): a one-rank communicator must leave the unsharded solve's bits."""
    s = scene.make_scene(**PROC_SCENE)
    ref = _single(s, [([0.9], 16.0)] * 3)
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER)
    for attempt in range(2):
        sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
        p = subprocess.run([sys.executable, str(script), str(port)], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        if p.returncode == 0:
            break
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert out["sha"] == _sha(ref["poses"]) and out["outliers"] == ref["outliers"] and out["calls"] > 0
