"""GPU tests (-m gpu) of bench.py's contract: ONE JSON line from rank 0, the headline fields, the roofline / cpu-baseline objects,
the extras of every BASELINE configuration at N = 1, and -- launched exactly as the driver launches a multi-GPU run -- the weak
(replica) record together with the strong-scaling record of the windowed 8k-pose map from two ranks (sharing the one GPU of the
test box: RCCL cannot form a two-rank communicator on one device, so the control plane and the pose-block exchange fall back to gloo;
on an 8-GPU node the same command uses RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}"
    return json.loads(lines[0])            # strict JSON: a NaN in the line would fail here


def test_single_gpu_line_carries_every_configuration():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["unit"] == "LM iterations/s" and d["scaling"] == "weak" and d["dtype"] == "f64"
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert d["trials_per_iteration"] == 1.0 and 1.2 < d["final_reproj_rmse_px"] < 2.0
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0.2 < r["frac"] < 1.0 and r["launches"] == 5 and r["ms_per_launch"] < d["ms_per_step"]
    h = d["roofline_hbm"]
    assert h["bound"] == "hbm" and set(h["stages"]) == {"linearize", "schur_build", "backsubst_and_trial_error"} and 0 < h["frac"] < 1
    # what lies outside the measured stages of a step (host turn-around, the outlier pass, the reductions) stays small
    # (0.04-0.06 ms measured; a 5-step window on a freshly woken GPU has shown 0.19 once: such a window is measured again, longer, and must hold then)
    over = d["ms_per_step"] - r["ms_per_launch"] - h["ms"]
    if over >= 0.12:
        p2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-extras"],
                            capture_output=True, text=True, timeout=900)
        assert p2.returncode == 0, p2.stderr[-3000:]
        d2 = _line(p2.stdout)
        over = d2["ms_per_step"] - d2["roofline"]["ms_per_launch"] - d2["roofline_hbm"]["ms"]
    assert over < 0.12, over
    e = d["extra"]
    for k in ("config2", "config2_cpu_baseline", "config3", "config3_cpu_baseline", "config4_end_to_end", "cold_start", "skyline_solve", "sustained", "concurrent_handles", "config5_windowed_1gpu"):
        assert k in e and "error" not in e[k], (k, e.get(k))
    assert e["config3"]["one_iteration_bundler_create_to_destroy"]["total_ms"] < 1.5      # 2.1 ms with the host structure build
    assert e["config4_end_to_end"]["one_iteration_bundler"]["first_step_ms_structure_build_plus_one_iteration"] < 8.0
    cs = e["cold_start"]
    assert cs["global"]["fresh_process"]["task_graph_size"] and cs["global"]["warm_process_new_size"]["tile_columns"] == 57 and cs["global2k"]["fresh_process"]["tile_columns"] == 94
    sk = e["skyline_solve"]
    assert sk["same_mse_to_the_bit"] and sk["skyline"]["factor_and_solves_ms"] < sk["dense"]["factor_and_solves_ms"] and "error" not in sk["global2k_skyline"]
    assert set(e["config2"]["batches"]) == {"1", "64", "1024"}
    assert e["sustained"]["seeded_lambda_5e6"]["timed_seconds"] >= 2.0 and e["sustained"]["seeded_lambda_5e6"]["trials_per_iteration"] == 1.0
    s = d["strong_scaling"]
    assert s["scaling"] == "strong" and s["n_gpus"] == 1 and s["windows_per_rank"] == 8 and s["value"] > 0


def test_two_ranks_under_torchrun_give_the_weak_and_the_strong_record():
    env = dict(os.environ, MAGE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2"],
                       capture_output=True, text=True, timeout=1200, env=env)
    assert p.returncode == 0, "\n".join(l for l in p.stderr.splitlines() if not l.startswith(("W0", "E0", "[Gloo]")))[-6000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "replica x2" in d["config"]["parallelism"] and d["config"]["control_plane"].startswith("gloo")
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]        # whole-job aggregate: both ranks' steps over the slower rank's time
    assert "cpu_baseline" not in d and "extra" not in d                           # rank 0 at N = 1 only
    s = d["strong_scaling"]
    assert "error" not in s, s
    assert s["scaling"] == "strong" and s["n_gpus"] == 2 and s["windows_per_rank"] == 4 and s["value"] > 0
    assert s["mse_of_rank0_windows"][-1] < s["mse_of_rank0_windows"][0]
    assert "gloo" in s["exchange"]


def test_eight_ranks_on_the_one_gpu_end_without_deadlock():
    """The command the driver runs on an 8-GPU node, with the eight ranks sharing this box's one GPU (gloo carries the barriers and the
    pose block): proves the world-8 rank arithmetic -- eight independent sub-maps in `value`, ONE window per rank in `strong_scaling`, every
    barrier and all-reduce matched -- without the node.  RCCL itself needs a GPU per rank and is not exercised here."""
    env = dict(os.environ, MAGE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=2400, env=env)
    assert p.returncode == 0, "\n".join(l for l in p.stderr.splitlines() if not l.startswith(("W0", "E0", "[Gloo]")))[-6000:]
    d = _line(p.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and "replica x8" in d["config"]["parallelism"]
    assert abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    s = d["strong_scaling"]
    assert "error" not in s, s
    assert s["n_gpus"] == 8 and s["windows_per_rank"] == 1 and s["value"] > 0 and len(s["mse_of_rank0_windows"]) >= 3
