"""CPU test of the N > 1 path: two processes over gloo exercise exactly the helpers bench.py uses for multi-GPU runs
(rank discovery from the environment, sub-map ownership, barrier, max/sum reductions of the timing statistics)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import sys, json
    sys.path.insert(0, %r)
    from mageslam_amd import dist as D, scene
    info = D.rank_info()
    dist = D.init("gloo", info)
    assert dist is not None and dist.get_world_size() == 2
    seed = D.submap_seed(0x5EED0001, info.rank)
    s = scene.make_scene(n_cams=6, n_pts=40, n_obs=240, seed=seed)
    owned = D.assign_submaps(5, info.rank, info.world)
    dist.barrier()
    el, n, rmse = D.reduce_stats(dist, 1.0 + info.rank, 10 * (info.rank + 1), 0.5 + info.rank)
    print(json.dumps(dict(rank=info.rank, seed=seed, owned=owned, el=el, n=n, rmse=rmse, first_uv=float(s.obs_uv[0, 0]))))
    dist.barrier(); dist.destroy_process_group()
""") % ROOT


def _free_port():
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    return port


def test_two_rank_gloo_roundtrip(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    for attempt in range(2):            # the port is free when probed, not reserved: one retry if something else took it meanwhile
        port = _free_port()
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        res = [p.communicate(timeout=180) + (p.returncode,) for p in procs]
        if all(rc == 0 for _, _, rc in res) or attempt == 1:
            break
    outs = []
    for o, e, rc in res:
        assert rc == 0, e[-2000:]
        outs.append(__import__("json").loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert outs[0]["owned"] == [0, 2, 4] and outs[1]["owned"] == [1, 3]
    assert outs[0]["seed"] != outs[1]["seed"] and outs[0]["first_uv"] != outs[1]["first_uv"]     # distinct sub-maps
    for d in outs:                                                                               # every rank sees the same reduced stats
        assert d["el"] == 2.0 and d["n"] == 30 and d["rmse"] == 1.5


FALLBACK_WORKER = textwrap.dedent("""
    import sys, json
    sys.path.insert(0, %r)
    from mageslam_amd import dist as D
    info = D.rank_info()
    dist = D.init("nccl", info, device_index=0, timeout_s=60)        # no GPU here: RCCL cannot come up
    assert dist is not None and D.init.backend == "gloo" and D.stats_device(0) == "cpu"
    dist.barrier()
    el, n, rmse = D.reduce_stats(dist, 1.0 + info.rank, 7, 0.25 * (info.rank + 1), device=D.stats_device(0))
    print(json.dumps(dict(rank=info.rank, el=el, n=n, rmse=rmse)))
    dist.barrier(); dist.destroy_process_group()
""") % ROOT


def test_rccl_bring_up_failure_falls_back_to_gloo_under_torchrun(tmp_path):
    """bench.py's control plane must survive a node whose RCCL does not initialise: launched exactly as the driver does
    (python -m torch.distributed.run, agent-hosted store), a failed "nccl" bring-up ends on gloo over the same store."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without GPUs so that RCCL bring-up fails")
    script = tmp_path / "worker.py"
    script.write_text(FALLBACK_WORKER)
    for attempt in range(2):
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(_free_port()), str(script)], capture_output=True, text=True, timeout=300)
        if p.returncode == 0:
            break
    assert p.returncode == 0, p.stderr[-3000:]
    # both ranks print to the launcher's stdout: their lines can arrive glued together, so take the objects, not the lines
    outs = [__import__("json").loads(o) for o in __import__("re").findall(r"\{[^{}]*\}", p.stdout)]
    assert len(outs) == 2 and all(d["el"] == 2.0 and d["n"] == 14 and d["rmse"] == 0.5 for d in outs)
    assert "falling back to gloo" in p.stderr
