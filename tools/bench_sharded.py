"""Landmark-sharded solve of one map (mageslam_amd/sharded.py) -- what the exchange costs.

  python tools/bench_sharded.py --ranks 2                 all ranks as threads of this process on ONE GPU (ThreadGroup): the
                                                          arithmetic of every rank plus a device-local sum, no xGMI -- a
                                                          functional run and the cost of the pack / unpack / redundant
                                                          factorisation, not a scaling number
  torchrun --nproc-per-node N tools/bench_sharded.py      one process per GPU, RCCL all-reduce on the packed system (TorchGroup)

Prints one JSON line: LM iterations/s of the ONE map, bytes exchanged per trial, final RMSE, and the single-handle rate beside it.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="global")
    ap.add_argument("--ranks", type=int, default=2, help="threads mode only")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pts", type=int, default=0, help="override the configuration's number of map points")
    ap.add_argument("--obs", type=int, default=0, help="override the configuration's number of observations")
    a = ap.parse_args()
    from mageslam_amd import dist as mdist, scene, sharded
    from mageslam_amd.bundler import BundlerLib, load_scene

    def bulk(b, s):
        load_scene(b, s, bulk=True)

    over = {}
    if a.pts:
        over["n_pts"] = a.pts
    if a.obs:
        over["n_obs"] = a.obs
    s = scene.make_config(a.config, **over)
    info = mdist.rank_info()
    out = {"config": a.config, "n_cams": s.n_cams, "n_pts": s.n_pts, "n_obs": s.n_obs}
    if info.world > 1:
        import torch
        torch.cuda.set_device(info.local_rank)
        d = mdist.init("nccl", info, info.local_rank)
        group = sharded.TorchGroup(d, info.local_rank)
        sb = sharded.ShardedBundler(s, info.rank, info.world, lambda: BundlerLib(False, info.local_rank), bulk, group.callback())
        for _ in range(a.warmup):
            sb.StepBundleAdjustment([0.9], 1e30, [])
        d.barrier(); torch.cuda.synchronize()
        c0, t0 = group.doubles, time.perf_counter()
        trials = 0
        for _ in range(a.steps):
            mse = sb.StepBundleAdjustment([0.9], 1e30, [])
            trials += sum(t["trials"] for t in sb.trace())
        d.barrier(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if info.rank == 0:
            out.update(mode="rccl", n_gpus=info.world, lm_iterations_per_s=a.steps / dt, ms_per_iteration=1e3 * dt / a.steps,
                       trials=trials, exchanged_MB_per_trial=8e-6 * (group.doubles - c0) / max(trials, 1), rmse_px=float(np.sqrt(mse)))
            print(json.dumps(out))
        d.destroy_process_group()
        return
    calls = [([0.9], 1e30)] * (a.warmup + a.steps)
    # single handle beside it
    g = BundlerLib(False)
    bulk(g, s)
    for _ in range(a.warmup):
        g.StepBundleAdjustment([0.9], 1e30, [])
    t0 = time.perf_counter()
    for _ in range(a.steps):
        mse1 = g.StepBundleAdjustment([0.9], 1e30, [])
    t_single = (time.perf_counter() - t0) / a.steps
    g.close()
    import threading
    group = sharded.ThreadGroup(a.ranks)
    owner = sharded.partition_landmarks(s.obs_pt, s.n_pts, a.ranks)
    shards = [sharded.ShardedBundler(s, r, a.ranks, BundlerLib, bulk, group.callback(r), owner) for r in range(a.ranks)]
    gate = threading.Barrier(a.ranks + 1)
    mse, trials = [0.0] * a.ranks, [0] * a.ranks

    def run(r):
        for _ in range(a.warmup):
            shards[r].StepBundleAdjustment([0.9], 1e30, [])
        gate.wait(); gate.wait()
        for _ in range(a.steps):
            mse[r] = shards[r].StepBundleAdjustment([0.9], 1e30, [])
            trials[r] += sum(t["trials"] for t in shards[r].trace())
        gate.wait()

    th = [threading.Thread(target=run, args=(r,)) for r in range(a.ranks)]
    for t in th:
        t.start()
    gate.wait()
    d0, t0 = group.doubles, time.perf_counter()
    gate.wait(); gate.wait()
    dt = time.perf_counter() - t0
    for t in th:
        t.join()
    out.update(mode="threads-on-one-gpu", ranks=a.ranks, ms_per_iteration=1e3 * dt / a.steps, trials=trials[0],
               exchanged_MB_per_trial=8e-6 * (group.doubles - d0) / max(trials[0], 1), rmse_px=float(np.sqrt(mse[0])),
               single_handle_ms_per_iteration=1e3 * t_single, single_handle_rmse_px=float(np.sqrt(mse1)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
