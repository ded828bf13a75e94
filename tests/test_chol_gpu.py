"""GPU tests (-m gpu) of the dense solver alone (mage_debug_dense_solve): the hand-written tiled Cholesky against LAPACK on
the host (scipy) and, where torch's ROCm build offers it, rocSOLVER through torch.linalg on the same GPU."""
import ctypes as C

import numpy as np
import pytest

from mageslam_amd.bundler import check, lib

pytestmark = pytest.mark.gpu


def dense_solve(A, b):
    L = lib()
    L.mage_debug_dense_solve.argtypes = [C.c_int, C.c_int, np.ctypeslib.ndpointer(np.float64, flags="F_CONTIGUOUS"),
                                         np.ctypeslib.ndpointer(np.float64), np.ctypeslib.ndpointer(np.float64), C.POINTER(C.c_int)]
    n = len(b)
    x = np.zeros(n); ok = C.c_int(-1)
    check(L.mage_debug_dense_solve(-1, n, np.asfortranarray(A), np.ascontiguousarray(b), x, C.byref(ok)))
    return x, ok.value


def spd(n, seed, cond_shift=1.0):
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n + 8))
    return M @ M.T / n + cond_shift * np.eye(n), rng.standard_normal(n)


@pytest.mark.parametrize("n", [1, 6, 17, 127, 128, 129, 300, 500, 1100, 1250, 1408, 2999])      # 1 .. 24 tile columns, odd and even (the backward solve pairs two workgroups per column)
def test_solution_matches_lapack(n):
    """Orders around the tile size (one tile, exactly one, one row more), a few tiles (split diagonal update, quartered
    rounds), and a size that needs every role of the update kernel."""
    import scipy.linalg
    A, b = spd(n, 100 + n)
    x, ok = dense_solve(A, b)
    assert ok == 1
    ref = scipy.linalg.cho_solve(scipy.linalg.cho_factor(A, lower=True), b)
    assert np.linalg.norm(x - ref) <= 1e-11 * np.linalg.norm(ref)
    assert np.linalg.norm(A @ x - b) <= 1e-12 * (np.linalg.norm(A) * np.linalg.norm(x) + np.linalg.norm(b))


def test_ill_conditioned_system_has_a_backward_stable_residual():
    A, b = spd(900, 7, cond_shift=1e-9)                                  # condition number ~1e10
    x, ok = dense_solve(A, b)
    assert ok == 1
    assert np.linalg.norm(A @ x - b) <= 1e-11 * (np.linalg.norm(A) * np.linalg.norm(x) + np.linalg.norm(b))


def test_indefinite_and_semidefinite_matrices_are_reported():
    A, b = spd(300, 3)
    A[150, 150] = -1.0                                                   # a negative pivot appears in the second tile
    assert dense_solve(A, b)[1] == 0
    Z = np.zeros((140, 140)); Z[:139, :139] = spd(139, 4)[0]             # exact zero pivot in the last position
    assert dense_solve(Z, np.ones(140))[1] == 0


def test_solution_matches_rocsolver_via_torch(tmp_path):
    """torch.linalg on the GPU is rocSOLVER / hipSOLVER.  torch must initialise HIP itself, so the reference solve runs in its
    own process and hands the solution back through a file."""
    import subprocess
    import sys
    A, b = spd(1408, 21)
    x, ok = dense_solve(A, b)
    assert ok == 1
    np.savez(tmp_path / "sys.npz", A=A, b=b)
    code = ("import numpy as np, torch, sys\n"
            "if not torch.cuda.is_available(): sys.exit(3)\n"
            "z = np.load(sys.argv[1]); A = torch.from_numpy(z['A']).cuda(); b = torch.from_numpy(z['b']).cuda()\n"
            "x = torch.cholesky_solve(b[:, None], torch.linalg.cholesky(A))[:, 0].cpu().numpy()\n"
            "np.save(sys.argv[2], x)\n")
    p = subprocess.run([sys.executable, "-c", code, str(tmp_path / "sys.npz"), str(tmp_path / "x.npy")], capture_output=True, text=True, timeout=600)
    if p.returncode == 3:
        pytest.skip("no GPU visible to torch")
    assert p.returncode == 0, p.stderr[-2000:]
    ref = np.load(tmp_path / "x.npy")
    assert np.linalg.norm(x - ref) <= 1e-11 * np.linalg.norm(ref)


def test_schedules_are_bit_identical(tmp_path):
    """The same factorisation under its schedules -- ONE persistent launch over the task list (default from 8 tile columns on; forced
    from 2 with MAGE_CHOL_DAG_MIN_TILES), with panels fused per task (MAGE_CHOL_DAG_FUSE=8, default) or one panel per task, column by
    column with the panel solve merged into the update launch (MAGE_CHOL_COLUMN_LAUNCHES=1) or as launches of its own
    (+ MAGE_CHOL_NO_MERGED_TRSM=1, what a process falls back to after a stalled hand-off): every element accumulates its panel columns
    in the same order into an accumulator that starts as the element, so the solution is the same to the bit.  (The switches are read
    once per process: children.)"""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import numpy as np
        import test_chol_gpu as T
        out = []
        for n in (300, 700, 2999, 4500):      # 3, 6, 24, 36 tile columns
            A, b = T.spd(n, 900 + n)
            x, ok = T.dense_solve(A, b)
            assert ok == 1
            out.append(x)
        np.save(sys.argv[1], np.concatenate(out))
    """) % (root, os.path.join(root, "tests"))
    res = {}
    for tag, env in (("task_graph", {}), ("task_graph_every_size", {"MAGE_CHOL_DAG_MIN_TILES": "2"}), ("task_graph_one_panel_per_task", {"MAGE_CHOL_DAG_MIN_TILES": "2", "MAGE_CHOL_DAG_FUSE": "1"}),
                     ("columns_merged", {"MAGE_CHOL_COLUMN_LAUNCHES": "1"}), ("columns_separate_panels", {"MAGE_CHOL_COLUMN_LAUNCHES": "1", "MAGE_CHOL_NO_MERGED_TRSM": "1"})):
        f = str(tmp_path / (tag + ".npy"))
        p = subprocess.run([sys.executable, "-c", code, f], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]
        res[tag] = np.load(f)
    for tag, x in res.items():
        assert np.array_equal(x, res["task_graph"]), tag


def test_a_stalled_task_graph_launch_is_rerun_column_by_column(tmp_path):
    """What a deployment sees when a bounded wait of the task-graph launch runs out (several processes oversubscribing the GPU): the launch
    aborts with *stall = 4, the LM trial is run again from the untouched linearisation by the column-by-column launches -- same bits -- and
    the process stays with them.  The stall is injected (mage_debug_chol_inject_stall); a child process, because the switch is for life."""
    import json, os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, json
        sys.path.insert(0, %r)
        import numpy as np
        from mageslam_amd import scene
        from mageslam_amd.bundler import BundlerLib, load_scene, lib
        inject = int(sys.argv[2])
        s = scene.make_scene(n_cams=200, n_pts=6000, n_obs=60000, seed=0x5EED0B12, outlier_frac=0.01)          # 1200 rows: 10 tile columns
        b = BundlerLib(False); load_scene(b, s, bulk=True)
        outs = []
        o = []; outs.append([float(b.StepBundleAdjustment([1.8], 25.0, o))] + sorted(o))
        if inject: lib().mage_debug_chol_inject_stall(1)
        for hub, thr in [([0.9, 0.9], 16.0), ([0.9], 9.0)]:
            o = []; outs.append([float(b.StepBundleAdjustment(hub, thr, o))] + sorted(o))
        p = b.profile()
        np.save(sys.argv[1], np.concatenate([b.poses_f64().ravel(), b.points_f64().ravel()]))
        print("RESULT " + json.dumps(dict(outs=outs, rerun=int(p.trials_rerun_after_stall), fallback=int(p.fallback_to_separate_launches))))
    """) % root
    res = {}
    for tag, inject in (("clean", 0), ("stalled", 1)):
        f = str(tmp_path / (tag + ".npy"))
        p = subprocess.run([sys.executable, "-c", code, f, str(inject)], capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]
        res[tag] = (np.load(f), json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:]))
    assert res["clean"][1]["rerun"] == 0 and res["clean"][1]["fallback"] == 0
    assert res["stalled"][1]["rerun"] == 1 and res["stalled"][1]["fallback"] == 1
    assert res["stalled"][1]["outs"] == res["clean"][1]["outs"]
    assert np.array_equal(res["stalled"][0], res["clean"][0])


def test_first_factorisations_of_a_new_size_fall_back_and_agree():
    """The asynchronous default (no MAGE_CHOL_DAG_SYNC_BUILD): in a fresh process the first step of a new map size does not wait for the
    task lists of the dense solve (tools/cold_start.py: a worker thread builds them, the factorisation goes column by column meanwhile),
    and the numbers are those of a process that waited -- the schedules give the same bits."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = {}
    for tag, sync in (("async", False), ("sync", True)):
        env = dict(os.environ)
        env.pop("MAGE_CHOL_DAG_SYNC_BUILD", None)
        if sync:
            env["MAGE_CHOL_DAG_SYNC_BUILD"] = "1"
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "cold_start.py"), "--workload", "global"], capture_output=True, text=True, timeout=900, env=env)
        assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]
        runs[tag] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])["fresh_process"]
    a, s = runs["async"], runs["sync"]
    assert a["task_graph_size"] and a["tile_columns"] == 47
    assert a["mse_first_second"] == s["mse_first_second"]



def dense_solve_skyline(A, b, env):
    L = lib()
    L.mage_debug_dense_solve_skyline.argtypes = [C.c_int, C.c_int, np.ctypeslib.ndpointer(np.float64, flags="F_CONTIGUOUS"), np.ctypeslib.ndpointer(np.float64),
                                                 np.ctypeslib.ndpointer(np.float64), C.POINTER(C.c_int), np.ctypeslib.ndpointer(np.int32)]
    n = len(b)
    x = np.zeros(n); ok = C.c_int(-1)
    check(L.mage_debug_dense_solve_skyline(-1, n, np.asfortranarray(A), np.ascontiguousarray(b), x, C.byref(ok), np.ascontiguousarray(env, dtype=np.int32)))
    return x, ok.value


@pytest.mark.parametrize("n,kind", [(1100, "band"), (2999, "band"), (2999, "ragged"), (4500, "arrow"), (1408, "dense")])
def test_skyline_schedule_gives_the_dense_solution_to_the_bit(n, kind):
    """mage_ba_use_skyline's solve: the task-graph schedule built from the matrix's skyline by tile rows never touches a tile left of it --
    tiles that are zero and stay zero in the factor -- so the solution has the bits of the dense schedule's.  Band, ragged (a different
    first tile per row), arrow (a few late rows reach back to column 0: the loop-closure shape) and the dense envelope itself."""
    rng = np.random.default_rng(7000 + n)
    nt = (n + 127) // 128
    if kind == "band":
        env = np.maximum(0, np.arange(nt) - 2)
    elif kind == "ragged":
        env = np.array([int(rng.integers(max(0, i - 4), i + 1)) for i in range(nt)])
    elif kind == "arrow":
        env = np.maximum(0, np.arange(nt) - 1); env[-3:] = 0
    else:
        env = np.zeros(nt, dtype=int)
    env[0] = 0
    # an SPD matrix whose non-zeros lie inside the skyline: a masked random matrix made diagonally dominant
    M = rng.standard_normal((n, n)) * 0.05
    M = np.tril(M)
    rows_tile = np.arange(n) // 128
    first_col = env[rows_tile] * 128
    M[np.arange(n)[None, :] < first_col[:, None]] = 0.0
    A = M + M.T
    A[np.diag_indices(n)] = np.abs(A).sum(axis=1) + 1.0
    b = rng.standard_normal(n)
    x_dense, ok_d = dense_solve(A, b)
    x_sky, ok_s = dense_solve_skyline(A, b, env)
    assert ok_d == 1 and ok_s == 1
    assert np.array_equal(x_dense, x_sky)
    assert np.linalg.norm(A @ x_sky - b) <= 1e-12 * np.linalg.norm(b) * n


def test_bundle_adjustment_with_the_skyline_solve_is_bit_identical():
    """A 300-camera trajectory map (1 800 rows: 15 tile columns, block-banded reduced system) stepped with and without
    mage_ba_use_skyline: the same errors, outlier lists and state to the bit."""
    from mageslam_amd import scene
    from mageslam_amd.bundler import BundlerLib, load_scene
    s = scene.make_scene(n_cams=300, n_pts=9000, n_obs=90000, seed=0x5EED0B31, outlier_frac=0.01)
    res = []
    for sky in (False, True):
        b = BundlerLib(False)
        b.use_skyline(sky)
        load_scene(b, s, bulk=True)
        outs = []
        for hub, thr in [([1.8], 25.0), ([0.9, 0.9], 16.0), ([0.9], 9.0)]:
            o = []
            outs.append((float(b.StepBundleAdjustment(hub, thr, o)), sorted(o)))
        res.append((outs, b.poses_f64().copy(), b.points_f64().copy()))
        b.close()
    assert res[0][0] == res[1][0]
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
