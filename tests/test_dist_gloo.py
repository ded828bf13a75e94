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


def test_two_rank_gloo_roundtrip(tmp_path):
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(__import__("json").loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert outs[0]["owned"] == [0, 2, 4] and outs[1]["owned"] == [1, 3]
    assert outs[0]["seed"] != outs[1]["seed"] and outs[0]["first_uv"] != outs[1]["first_uv"]     # distinct sub-maps
    for d in outs:                                                                               # every rank sees the same reduced stats
        assert d["el"] == 2.0 and d["n"] == 30 and d["rmse"] == 1.5
