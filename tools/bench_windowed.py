"""BASELINE.json configs[4], second form: ONE 8k-pose map sharded by keyframe window (mageslam_amd/windowed.py).

    python tools/bench_windowed.py [--poses 8000 --windows 8 --overlap 10 --iters 10]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/bench_windowed.py ...

Every rank owns windows/N consecutive windows (strong scaling: the map is fixed); one outer iteration = one LM iteration in
every window + the all-reduce of the (poses x 8) float64 pose block in HBM (RCCL on a multi-GPU node) + the halo re-seed.  Prints one JSON
line from rank 0: outer iterations/s, the error trajectory, bytes exchanged.
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
    ap.add_argument("--poses", type=int, default=8000)
    ap.add_argument("--windows", type=int, default=8)
    ap.add_argument("--overlap", type=int, default=10)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--threads", type=int, default=4, help="windows of a rank stepped concurrently")
    ap.add_argument("--cpp", action="store_true", help="the C++ driver (include/mage_window.h) instead of mageslam_amd/windowed.py; one rank")
    a = ap.parse_args()
    import torch
    from mageslam_amd import dist as D, scene
    from mageslam_amd.bundler import BundlerLib, load_scene
    from mageslam_amd.windowed import WindowedMap
    info = D.rank_info()
    device = info.local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device)
    dist = D.init(os.environ.get("MAGE_DIST_BACKEND", "nccl"), info, device_index=device)
    t0 = time.perf_counter()
    s = scene.make_scene(n_cams=a.poses, n_pts=100 * a.poses, n_obs=1000 * a.poses, seed=0x5EED0008)
    t1 = time.perf_counter()
    if a.cpp:
        from mageslam_amd.wmap import WindowMap
        assert info.world == 1, "the C++ driver's multi-rank form is tools/windowed_rccl.cpp"
        m = WindowMap(s, a.windows, overlap=a.overlap, device=device, threads=a.threads)
        m.mine = list(range(a.windows))
    else:
        m = WindowedMap(s, a.windows, lambda: BundlerLib(False, device=device), lambda b, sc: load_scene(b, sc, bulk=True),
                        rank=info.rank, world=info.world, dist=dist, overlap=a.overlap, exchange_device=D.stats_device(device), threads=a.threads,
                        device=device)
    t2 = time.perf_counter()
    errs = [m.outer_iteration(1.8) for _ in range(a.warmup)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    errs += [m.outer_iteration(1.8) for _ in range(a.iters)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    if info.rank == 0:
        if a.cpp:
            sizes = [(i["own"], i["cams"], i["obs"]) for i in (m.window_info(w) for w in m.mine)]
        else:
            sizes = [(len(m.windows[w].own), m.windows[w].scene.n_cams, m.windows[w].scene.n_obs) for w in m.mine]
        print(json.dumps({
            "workload": f"one map of {a.poses} poses / {100 * a.poses} points / {1000 * a.poses} observations in {a.windows} windows, overlap {a.overlap}",
            "n_gpus": info.world, "scaling": "strong", "threads_per_rank": a.threads, "control_plane": D.init.backend or "none",
            "outer_iterations_per_s": a.iters / (t4 - t3), "ms_per_outer_iteration": 1e3 * (t4 - t3) / a.iters,
            "lm_window_iterations_per_s": a.iters * a.windows / (t4 - t3),
            "mse_rank0_windows": [round(float(e), 5) for e in errs],
            "driver": "C++ (mage_wmap_*, exchange in HBM)" if a.cpp else "python (windowed.py, pose block in HBM)",
            "exchange_bytes_per_iteration": a.poses * 8 * 8,
            "rank0_windows_own_cams_obs": sizes,
            "scene_s": round(t1 - t0, 1), "cut_and_load_s": round(t2 - t1, 1)}), flush=True)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
