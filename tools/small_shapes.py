"""The reference's real operating points (bench.py SMALL_SHAPES), device path and CPU oracle side by side, both from native callers:
tools/_bin/shim_small_shapes (C++ over include/BundlerLib.h) and oracle/_build/oracle_small_shapes (C over the oracle).
    python tools/small_shapes.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

if __name__ == "__main__":
    print(json.dumps({"hip": bench.extra_small_shapes(), "cpu_oracle": bench.cpu_baseline_small_shapes()}, indent=1))
