#!/usr/bin/env python3
"""Development driver: three StepBundleAdjustment calls on the 1k-pose map (seeded lambda), nothing else -- what the per-kernel counter
passes (tools/pmc_kernel.sh) and the ablation builds of the stage kernels are run over.  MAGE_LIB=<path> loads another build of the library.

    rocprofv3 --kernel-trace -d /tmp/p -o b -- python tools/schur_time.py ; python tools/rocpd_stats.py /tmp/p/.../b_results.db
"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mageslam_amd import _lib  # noqa: E402
if os.environ.get("MAGE_LIB"):
    _lib.LIB_PATH = os.environ["MAGE_LIB"]
from mageslam_amd import scene  # noqa: E402
from mageslam_amd.bundler import BundlerLib, load_scene  # noqa: E402

s = scene.make_config("global")
b = BundlerLib(False, device=0)
load_scene(b, s, bulk=True)
b.SetCurrentLambda(5e6)
out = []
for _ in range(3):
    try:
        b.StepBundleAdjustment([1.8], 1e30, out)
    except Exception as e:          # an ablation build may produce an indefinite system: the kernels have run all the same
        print("step:", e)
