import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from mageslam_amd import scene
from mageslam_amd.bundler import BundlerLib, load_scene
s = scene.make_config("local", outlier_frac=0.02)
def run(verbose=False):
    t0 = time.perf_counter()
    b = BundlerLib(False, device=0)
    t1 = time.perf_counter()
    load_scene(b, s, bulk=True)
    t2 = time.perf_counter()
    thr, o = 7.25, []
    ts = []
    for _ in range(10):
        ta = time.perf_counter(); b.StepBundleAdjustment([0.9], thr, o); thr *= 0.95 * 0.95; ts.append(time.perf_counter() - ta)
    t3 = time.perf_counter()
    b.close()
    t4 = time.perf_counter()
    return dict(create=t1 - t0, load=t2 - t1, steps=ts, close=t4 - t3, total=t4 - t0)
for _ in range(3): run()
rs = [run() for _ in range(30)]
avg = lambda k: 1e3 * np.mean([r[k] for r in rs])
print("ms: create %.3f load(set arrays) %.3f close %.3f total %.3f" % (avg("create"), avg("load"), avg("close"), avg("total")))
print("steps ms:", " ".join("%.3f" % (1e3 * np.mean([r["steps"][i] for r in rs])) for i in range(10)))
