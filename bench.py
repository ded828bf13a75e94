#!/usr/bin/env python3
"""bench.py -- LM iterations/s of the MI355X bundle-adjustment back-end on BASELINE.json's headline workload.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N = 1 : the 1k-pose / 100k-point / 1M-observation synthetic global BA (BASELINE.json configs[3], the
          configuration the metric is quoted on) on cuda:0.
  N > 1 : launched by torch.distributed.run, one rank per GPU; every rank owns an independent sub-map of
          the same shape (different seed) -- BASELINE.json configs[4], weak scaling, no data-path collective.
A "step" is one Levenberg-Marquardt iteration = one BundlerLib::StepBundleAdjustment call with a single
Huber width (all its damped trials, plus the reference's outlier-classification pass).  Inputs are
resident in HBM when the timed region starts (the problem is uploaded and the graph structure is
built during warm-up).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0         # MI355X HBM3E (/opt/skills/guides/MI355X_MICROARCH.md: ~8 TB/s peak, ~6.3 TB/s achievable)
F64_MFMA_PEAK_TFLOPS = 78.6   # v_mfma_f64_16x16x4_f64: 32 flop/clk/SIMD * 4 SIMD * 256 CU * 2.4 GHz (AMD MI355X spec)
HUBER = 1.8                   # BundleAdjustSettings.HuberWidth default (MageSettings.h:41-52)
# "for throughput runs seed lambda so that >= 95 % of iterations take exactly 1 trial" (report trials/iter).  With
# g2o's own start (tau * max diag = 500) the damping has decayed to ~2e-4 by iteration 19 and a handful of iterations then
# need a second trial; SetCurrentLambda(5e6) (BundlerLib.cpp:354-362) moves that phase past iteration 26.  Same minimum
# (RMSE 1.30294 px) either way; a second trial only ever ADDS work to an iteration.
LAMBDA_SEED = {"global": 5e6}
WORKLOADS = {
    "global": dict(n_cams=1000, n_pts=100000, n_obs=1000000, seed=0x5EED0004),
    "local": dict(n_cams=20, n_pts=5000, n_obs=50000, seed=0x5EED0003, fixed=(0, 1, 15, 16, 17, 18, 19)),
    "tiny": dict(n_cams=10, n_pts=200, n_obs=2000, seed=0x5EED0001),
}


def _oracle_iterations(args):
    """One replica of the CPU baseline: n LM iterations of the oracle on the workload's scene (seed offset per replica)."""
    workload, n_iter, replica = args
    import time as _t
    from mageslam_amd import scene
    from oracle.oracle import OracleBundler, load_scene_bulk
    kw = dict(WORKLOADS[workload])
    kw["seed"] = kw["seed"] + 0x100 * replica
    s = scene.make_scene(**kw)
    b = OracleBundler()
    load_scene_bulk(b, s)
    if workload in LAMBDA_SEED:
        b.SetCurrentLambda(LAMBDA_SEED[workload])
    out: list = []
    t0 = _t.perf_counter()
    for _ in range(n_iter):
        b.StepBundleAdjustment([HUBER], 1e30, out)
    return _t.perf_counter() - t0


def cpu_baseline(workload: str, min_iterations: int = 2, max_seconds: float = 45.0, replicas: int = 8) -> dict:
    """Times the CPU oracle (oracle/ba_oracle.c: same algorithm incl. the reference's unblocked pivoted LDLT, one thread like the
    reference's single-threaded g2o path) on the same scene for a bounded sample: at least `min_iterations` LM iterations on
    one core, then `replicas` independent sub-maps side by side, one per core (the multi-GPU comparison)."""
    el1 = _oracle_iterations((workload, 1, 0))
    n = max(min_iterations, min(50, int(10.0 / max(el1, 1e-9))))
    if n * el1 > max_seconds:
        n = max(min_iterations, int(max_seconds / el1))
    el = _oracle_iterations((workload, n, 0))
    res = {"value": n / el, "unit": "LM iterations/s", "cores": 1, "host_cores": os.cpu_count(), "kind": "port",
           "sample": f"{n} LM iteration(s) of the same {workload} scene from the same initial state, {el:.1f} s, "
                     f"oracle/ba_oracle.c (gcc -O3), 1 thread"}
    try:
        import multiprocessing as mp
        r = min(replicas, os.cpu_count() or 1)
        t0 = time.perf_counter()
        with mp.get_context("spawn").Pool(r) as pool:
            pool.map(_oracle_iterations, [(workload, 1, i) for i in range(r)])
        wall = time.perf_counter() - t0
        res["replicas"] = {"value": r / wall, "unit": "LM iterations/s", "cores": r,
                           "sample": f"{r} independent sub-maps, one oracle process per core, 1 LM iteration each, {wall:.1f} s wall "
                                     f"(scene generation included)"}
    except Exception as e:  # noqa: a report only
        res["replicas"] = {"value": None, "sample": f"failed: {e}"}
    return res


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="global", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from mageslam_amd import dist as D
    info = D.rank_info()
    rank, local_rank, world = info.rank, info.local_rank, info.world
    if args.gpus != world and world > 1:
        print(f"--gpus {args.gpus} != WORLD_SIZE {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X: torch.cuda.is_available() is False", file=sys.stderr)
        return 2
    n_dev = torch.cuda.device_count()
    device = local_rank % max(n_dev, 1)             # one GPU per rank on the node; the modulo only matters for the single-GPU plumbing test
    torch.cuda.set_device(device)
    # "nccl" is RCCL on ROCm.  MAGE_DIST_BACKEND=gloo lets the N > 1 plumbing be exercised on a box with fewer GPUs than ranks.
    dist = D.init(os.environ.get("MAGE_DIST_BACKEND", "nccl"), info, device_index=device)

    from mageslam_amd import scene
    from mageslam_amd.bundler import BundlerLib, load_scene

    kw = dict(WORKLOADS[args.workload])
    kw["seed"] = D.submap_seed(kw["seed"], rank)    # independent sub-map per rank (weak scaling, no data-path collective)
    s = scene.make_scene(**kw)
    b = BundlerLib(False, device=device)
    load_scene(b, s, bulk=True)
    if args.workload in LAMBDA_SEED:
        b.SetCurrentLambda(LAMBDA_SEED[args.workload])
    outl: list = []
    trials = 0
    for _ in range(args.warmup):                    # uploads the problem, builds the graph structure
        b.StepBundleAdjustment([HUBER], 1e30, outl)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    b.enable_profiling(True)
    barrier()
    t0 = time.perf_counter()
    mse = float("nan")
    for _ in range(args.steps):
        mse = b.StepBundleAdjustment([HUBER], 1e30, outl)
        trials += sum(t["trials"] for t in b.trace())
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed, total_steps, worst_rmse = D.reduce_stats(dist, elapsed, args.steps, float(np.sqrt(mse)), device=D.stats_device(device))
    prof = b.profile()

    if rank == 0:
        value = total_steps / elapsed
        nfac = max(int(prof.n_factorizations), 1)
        fac_ms = prof.factor_ms_total / nfac
        flops = prof.factor_flops_each
        achieved = flops / (fac_ms * 1e-3) / 1e12 if fac_ms > 0 else 0.0
        traffic = None
        try:   # HBM bytes per factorisation from the committed PMC passes (tools/pmc_to_traffic.py); same padded order only
            tj = json.load(open(os.path.join(ROOT, "profiles", "chol_traffic.json")))
            if int(tj.get("n_pad", -1)) == int(prof.padded_order):
                traffic = float(tj["bytes_per_factorization"])
        except Exception:
            traffic = None
        line = {
            "metric": "LM iterations/sec + final reproj RMSE, 1k poses / 100k pts / 1M obs synthetic BA",
            "value": value, "unit": "LM iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "final_reproj_rmse_px": worst_rmse,
            "trials_per_iteration": trials / max(args.steps, 1),
            "config": {"workload": f"{args.workload}: {kw['n_cams']} poses / {kw['n_pts']} points / {kw['n_obs']} observations, "
                                   f"Huber {HUBER}, poses 0,1 fixed, lambda seed {LAMBDA_SEED.get(args.workload, 'g2o default')}, "
                                   f"one independent sub-map per GPU",
                       "parallelism": f"replica x{world} (independent sub-maps)",
                       "control_plane": (D.init.backend or "none") + (" (RCCL)" if D.init.backend == "nccl" else "")},
            "roofline": {
                "kernel": "dense reduced-camera Cholesky (k_potrf_diag + k_trsm_panel + k_syrk_update2 / k_syrk_update[f64 MFMA] + k_bsolve_persist), "
                          "HIP-event span per factorisation on the solver stream",
                "bound": "mfma", "achieved": achieved, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / F64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                "traffic_source": "committed profile (profiles/chol_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over "
                                  "tools/_bin/chol_test at the same padded order, (2*FETCH_SIZE + WRITE_SIZE)*1024 per factorisation; regenerate with "
                                  "tools/regen_profiles.sh) -- NOT measured in this run",
                "flops_per_launch": flops, "ms_per_launch": fac_ms, "launches": int(prof.n_factorizations),
                "system_order": int(prof.system_order), "padded_order": int(prof.padded_order),
                "schur_ms_per_launch": prof.schur_ms_total / max(int(prof.schur_launches), 1),
            },
        }
        # the HBM-bound stages of the same iterations: linearise (once per iteration), Schur build and back-substitution + trial
        # evaluation (once per trial); algorithmic bytes from the handle (every array a stage reads or writes counted once)
        n_lin, n_sch, n_upd = max(int(prof.linearize_launches), 1), max(int(prof.schur_launches), 1), max(int(prof.update_launches), 1)
        stages = {
            "linearize": (prof.linearize_ms_total / n_lin, prof.linearize_bytes_each),
            "schur_build": (prof.schur_ms_total / n_sch, prof.schur_bytes_each),
            "backsubst_and_trial_error": (prof.update_ms_total / n_upd, prof.update_bytes_each),
        }
        tot_ms = sum(ms for ms, _ in stages.values())
        tot_b = sum(by for _, by in stages.values())
        line["roofline_hbm"] = {
            "kernels": "k_small_linearize in its large-problem form (residuals + landmark side with 8 lanes per landmark + camera side, one launch) + k_reduce_sum | S zero-fill + k_lm_invert + k_schur_block (its diagonal blocks also give the reduced rhs) | k_pose_update + k_backsub (which also evaluates the trial's residuals) + k_reduce_sum; "
                       "HIP-event spans on the solver stream, per LM iteration with one trial",
            "bound": "hbm", "achieved": tot_b / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (tot_b / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if tot_ms > 0 else 0.0,
            "algorithmic_bytes": tot_b, "ms": tot_ms,
            "stages": {k: {"ms": ms, "algorithmic_bytes": by, "GB/s": by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                           "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0} for k, (ms, by) in stages.items()},
            "traffic": None,
        }
        try:   # HBM bytes per LM iteration of the three stages from the committed PMC passes (tools/pmc_stage_traffic.py)
            sj = json.load(open(os.path.join(ROOT, "profiles", "ba_stage_traffic.json")))
            if args.workload == "global":
                hb = line["roofline_hbm"]
                hb["traffic"] = float(sj["total_hbm_bytes_per_iteration"])
                for k, v in sj["stages_hbm_bytes_per_iteration"].items():
                    if k in hb["stages"]:
                        hb["stages"][k]["traffic"] = float(v)
                hb["traffic_source"] = ("committed profile (profiles/ba_stage_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over "
                                        "bench.py on the 1k-pose map), not measured in this run")
        except Exception:  # noqa: BLE001 - the profile is optional
            pass
        if not args.no_cpu_baseline and world == 1:   # reported baseline: rank 0, N = 1 only
            try:
                line["cpu_baseline"] = cpu_baseline(args.workload)
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "LM iterations/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
