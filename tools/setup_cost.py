"""Host-side cost of handing a problem to the bundler: bulk setters, structure build (first step), and one steady-state step.

    MAGE_BA_TIMING=1 python tools/setup_cost.py [--workload global]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="global"); a = ap.parse_args()
    from mageslam_amd import scene
    from mageslam_amd.bundler import BundlerLib, load_scene
    s = scene.make_config(a.workload)
    BundlerLib(False)                                  # device / library initialisation outside the timings
    for rep in range(3):
        t0 = time.perf_counter()
        b = BundlerLib(False)
        t1 = time.perf_counter()
        load_scene(b, s, bulk=True)
        t2 = time.perf_counter()
        b.StepBundleAdjustment([1.8], 1e30, [])
        t3 = time.perf_counter()
        b.StepBundleAdjustment([1.8], 1e30, [])
        t4 = time.perf_counter()
        del b
        t5 = time.perf_counter()
        print(f"rep {rep}: create {1e3*(t1-t0):.2f} ms, bulk setters {1e3*(t2-t1):.2f} ms, first step {1e3*(t3-t2):.2f} ms, "
              f"second step {1e3*(t4-t3):.2f} ms, destroy {1e3*(t5-t4):.2f} ms", flush=True)


if __name__ == "__main__":
    main()
