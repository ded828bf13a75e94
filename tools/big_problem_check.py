import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from mageslam_amd import scene
from mageslam_amd.bundler import BundlerLib, load_scene
t0=time.time()
s = scene.make_scene(n_cams=2500, n_pts=150000, n_obs=1500000, seed=0x5EED0099)
print("scene", time.time()-t0)
b = BundlerLib(False); load_scene(b, s, bulk=True)
out=[]
for i in range(6):
    t0=time.time(); m=b.StepBundleAdjustment([1.8], 1e30, out); dt=time.time()-t0
    tr=b.trace()[0]
    print(i, "mse", m, "trials", tr["trials"], "chi", tr["chi_before"], "->", tr["chi_after"], "lam", tr["lam"], "ms", round(dt*1e3,1))
p=b.profile(); print("order", p.system_order, p.padded_order, "fac ms", p.factor_ms_total/max(p.n_factorizations,1))
