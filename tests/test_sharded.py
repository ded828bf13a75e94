"""CPU tests of the landmark-sharded map's host side (mageslam_amd/sharded.py, include/mage_ba.h): the partition the ranks compute
independently (a host-only C-ABI call), a rank's sub-problem, and -- two processes over gloo -- the all-reduce callback the
solver calls per trial, here on host memory.  What the callback feeds is tested on the GPU (tests/test_sharded_gpu.py)."""
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from mageslam_amd import scene, sharded

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n_ranks", [1, 2, 3, 8])
def test_partition_is_balanced_deterministic_and_complete(n_ranks):
    s = scene.make_scene(n_cams=30, n_pts=3000, n_obs=24000, seed=77)
    rng = np.random.default_rng(5)
    keep = rng.random(s.n_obs) < 0.7                            # ragged track lengths, a few points unobserved
    obs_pt = s.obs_pt[keep]
    owner = sharded.partition_landmarks(obs_pt, s.n_pts, n_ranks)
    assert owner.shape == (s.n_pts,) and owner.min() >= 0 and owner.max() < n_ranks
    assert np.array_equal(owner, sharded.partition_landmarks(obs_pt.copy(), s.n_pts, n_ranks))
    w = sharded.landmark_weights(obs_pt, s.n_pts)
    loads = np.bincount(owner, weights=w, minlength=n_ranks)
    assert loads.max() - loads.min() <= w.max()                # greedy longest-first: within one point of each other
    # the same table as the plain restatement of the rule
    import heapq
    heap = [(0, r) for r in range(n_ranks)]
    want = np.zeros(s.n_pts, np.int32)
    for p in np.lexsort((np.arange(s.n_pts), -w)):
        load, r = heapq.heappop(heap)
        want[p] = r
        heapq.heappush(heap, (load + int(w[p]), r))
    assert np.array_equal(owner, want)


def test_partition_edge_cases():
    assert sharded.partition_landmarks(np.zeros(0, np.uint32), 0, 4).shape == (0,)
    assert np.array_equal(sharded.partition_landmarks(np.zeros(0, np.uint32), 3, 2), [0, 0, 0])      # unobserved points weigh nothing: the lightest rank stays rank 0
    with pytest.raises(Exception):
        sharded.partition_landmarks(np.array([5], np.uint32), 3, 2)                                   # point index out of range
    with pytest.raises(ValueError):
        sharded.partition_landmarks(np.zeros(0, np.uint32), 3, 0)


def test_shards_reassemble_the_map():
    s = scene.make_scene(n_cams=12, n_pts=600, n_obs=3600, seed=78, fixed=(0, 3))
    s.tethers = scene.make_tethers(s, n_dist=5, n_rot=4, n_xf=3, seed=79)
    n = 3
    owner = sharded.partition_landmarks(s.obs_pt, s.n_pts, n)
    seen_obs, seen_pts, n_teth = [], [], 0
    for r in range(n):
        sub, pts, obs = sharded.shard_scene(s, owner, r, n)
        assert sub.n_cams == s.n_cams and np.array_equal(sub.cam_fixed, s.cam_fixed) and sub.cam_t is s.cam_t       # every rank: all cameras
        assert np.array_equal(pts[sub.obs_pt], s.obs_pt[obs]) and np.array_equal(sub.obs_cam, s.obs_cam[obs])
        assert np.array_equal(sub.points, s.points[pts]) and np.array_equal(sub.obs_uv, s.obs_uv[obs])
        assert np.all(np.diff(obs) > 0)                                                                              # map order kept
        seen_obs.append(obs); seen_pts.append(pts)
        n_teth += len(sub.tethers.dist_d) + len(sub.tethers.rot_w) + len(sub.tethers.xf_w)
    assert np.array_equal(np.sort(np.concatenate(seen_obs)), np.arange(s.n_obs))
    assert np.array_equal(np.sort(np.concatenate(seen_pts)), np.arange(s.n_pts))
    assert n_teth == 12


WORKER = textwrap.dedent("""
    import sys, json, ctypes as C
    sys.path.insert(0, %r)
    import numpy as np
    from mageslam_amd import dist as D, scene, sharded
    info = D.rank_info()
    dist = D.init("gloo", info)
    s = scene.make_scene(n_cams=8, n_pts=400, n_obs=2400, seed=80)
    owner = sharded.partition_landmarks(s.obs_pt, s.n_pts, info.world)           # computed by each rank on its own
    sub, pts, obs = sharded.shard_scene(s, owner, info.rank, info.world)
    group = sharded.TorchGroup(dist, None)                                       # host memory: the plumbing of the callback
    cb = group.callback()
    n_pad = 256
    count = n_pad * (n_pad + 128) // 2 + n_pad                                   # the packed system of a 2-tile camera matrix
    buf = (np.arange(count, dtype=np.float64) + 1.0) * (info.rank + 1)
    assert cb(None, buf.ctypes.data, count, sharded.OP_SUM, None) == 0
    flag = np.array([float(info.rank == 1)])
    assert cb(None, flag.ctypes.data, 1, sharded.OP_MAX, None) == 0
    chk = np.array([float(np.sum(owner * np.arange(1, s.n_pts + 1)))])
    both = np.array([chk[0] * (1 if info.rank == 0 else -1)])
    assert cb(None, both.ctypes.data, 1, sharded.OP_SUM, None) == 0              # the two ranks' tables agree <=> 0
    print("RESULT " + json.dumps(dict(rank=info.rank, ok=bool(np.array_equal(buf, (np.arange(count) + 1.0) * 3.0)), flag=float(flag[0]),
                                      diff=float(both[0]), n_own=int(sub.n_pts), calls=group.calls, doubles=group.doubles)), flush=True)
    dist.barrier(); dist.destroy_process_group()
""") % ROOT


def test_two_ranks_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    for attempt in range(2):            # the port is free when probed, not reserved: one retry if something else took it meanwhile
        sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
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
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("RESULT ")][-1][7:]))
    outs.sort(key=lambda d: d["rank"])
    assert all(d["ok"] and d["flag"] == 1.0 and d["diff"] == 0.0 and d["calls"] == 3 for d in outs)
    assert outs[0]["n_own"] + outs[1]["n_own"] == 400
