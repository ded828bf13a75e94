#!/usr/bin/env python3
"""What the FIRST bundle adjustment of a new map size costs in a FRESH process -- the reference's call pattern after a loop closure
(Tasks/LoopClosureWorker.cpp:163-208; a bundler per optimisation, BundleAdjust.cpp:293, 348-351): create -> bulk set -> ONE
StepBundleAdjustment (structure build + one LM iteration) -> a second step -> destroy.  The dense solve's task lists for the new size are
built by a worker thread (chol_dag.hip: started at the structure build); until they are there the factorisation goes column by column.
Reported: every phase, how long the lists took on their thread, and how long a caller that WAITED for them would have waited after the
first step.  bench.py runs this in a child process per workload (extra.cold_start).

    python tools/cold_start.py [--workload global|global2k] [--device 0]
"""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

WORKLOADS = {"global": dict(n_cams=1000, n_pts=100000, n_obs=1000000, seed=0x5EED0004),
             "global1200": dict(n_cams=1200, n_pts=120000, n_obs=1200000, seed=0x5EED0024),      # (the map has grown: a NEW size in a process that is warm)
             "global2k": dict(n_cams=2000, n_pts=200000, n_obs=2000000, seed=0x5EED0014)}


def one(workload, device, L, skyline=False):
    from mageslam_amd import scene
    from mageslam_amd.bundler import BundlerLib, load_scene
    s = scene.make_scene(**WORKLOADS[workload])                  # (host work, not part of any phase)
    t0 = time.perf_counter()
    b = BundlerLib(False, device=device)
    if skyline:
        b.use_skyline(True)          # the dense solve skips the tiles left of the reduced system's skyline (same numbers)
    t1 = time.perf_counter()
    load_scene(b, s, bulk=True)
    b.SetCurrentLambda(5e6)
    t2 = time.perf_counter()
    out = []
    mse1 = b.StepBundleAdjustment([1.8], 1e30, out)
    t3 = time.perf_counter()
    p = b.profile()
    ms = C.c_double(0.0)
    # (the skyline's lists are a few hundred tasks, built in under a millisecond: nothing to wait for, and the dense lists are not wanted)
    dag = True if skyline else L.mage_debug_chol_wait_schedule(device, int(p.padded_order), C.byref(ms))
    t4 = time.perf_counter()
    mse2 = b.StepBundleAdjustment([1.8], 1e30, out)
    t5 = time.perf_counter()
    b.StepBundleAdjustment([1.8], 1e30, out)
    t6 = time.perf_counter()
    steady = None
    if skyline:          # the skyline's task lists take a millisecond to build: a few more steps are the steady state
        for _ in range(3):
            b.StepBundleAdjustment([1.8], 1e30, out)
        b.enable_profiling(2)
        ts = time.perf_counter()
        for _ in range(10):
            b.StepBundleAdjustment([1.8], 1e30, out)
        pp = b.profile()
        steady = {"ms_per_step": round(1e2 * (time.perf_counter() - ts), 4), "factor_and_solves_ms": round(pp.factor_ms_total / max(int(pp.n_factorizations), 1), 4)}
        b.enable_profiling(0)
    b.close()
    t7 = time.perf_counter()
    r = lambda x: round(1e3 * x, 3)
    return {"workload": workload, "padded_order": int(p.padded_order), "tile_columns": int(p.padded_order) // 128,
            "create_ms": r(t1 - t0), "bulk_set_ms": r(t2 - t1), "first_step_ms": r(t3 - t2),
            "task_lists_build_ms_on_worker_thread": round(ms.value, 2), "task_graph_size": bool(dag),
            "wait_for_task_lists_after_first_step_ms": r(t4 - t3),
            "second_step_ms": r(t5 - t4), "third_step_ms": r(t6 - t5), "destroy_ms": r(t7 - t6),
            "mse_first_second": [round(float(mse1), 6), round(float(mse2), 6)], **({"skyline_steady_state": steady} if steady else {})}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="global"); ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--then", default="", help="a second workload in the SAME process: a new map size in a warm process")
    ap.add_argument("--skyline", action="store_true", help="mage_ba_use_skyline on: + the steady state of the skyline solve")
    a = ap.parse_args()
    from mageslam_amd.bundler import lib
    L = lib()
    L.mage_debug_chol_wait_schedule.restype = C.c_int
    L.mage_debug_chol_wait_schedule.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double)]
    res = {"fresh_process": one(a.workload, a.device, L, a.skyline),
           "note": "the first step's factorisation runs column by column when the task lists are not there yet (same bits); the process's first asynchronous host-to-device copy (~6.5 ms of runtime set-up) is made by a worker thread of mage_ba_create (MAGE_BA_NO_WARMUP=1: inside the first step); "
                   "a fresh process also loads the code object and initialises the runtime inside its first step"}
    if a.then:
        res["warm_process_new_size"] = one(a.then, a.device, L)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
