"""Scratch comparison of the HIP bundler against the CPU oracle (prints, no asserts)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from mageslam_amd import scene
from mageslam_amd.bundler import BundlerLib, load_scene
from oracle.oracle import OracleBundler, load_scene as oload

def run(name, iters, huber, thr, outlier_frac=0.0, points_fixed=False, **kw):
    s = scene.make_config(name, outlier_frac=outlier_frac, **kw)
    g = BundlerLib(points_fixed); load_scene(g, s, bulk=True)
    o = OracleBundler(points_fixed); oload(o, s)
    og, oo = [], []
    for it in range(iters):
        t0 = time.time(); rg = g.StepBundleAdjustment([huber], thr, og); t1 = time.time()
        ro = o.StepBundleAdjustment([huber], thr, oo); t2 = time.time()
        tg, to = g.trace(), o.trace()
        if not tg or not to:
            print(it, 'no trace', tg, to); continue
        tg, to = tg[0], to[0]
        print(f"{name} it{it} mse gpu {rg:.7f} cpu {ro:.7f} trials {tg['trials']}/{to['trials']} "
              f"chi rel {abs(tg['chi_after']-to['chi_after'])/max(to['chi_after'],1e-300):.2e} "
              f"lam rel {abs(tg['lam']-to['lam'])/to['lam']:.2e} outl {len(og)}/{len(oo)} same {og==oo} "
              f"t_gpu {1e3*(t1-t0):.1f}ms t_cpu {1e3*(t2-t1):.1f}ms")
    Pg, Po = g.points_f64(), o.points_f64()
    Qg, Qo = g.poses_f64(), o.poses_f64()
    print(name, 'points max rel dev', np.abs(Pg-Po).max()/np.abs(Po).max(), 'pose t max abs dev', np.abs(Qg[:,4:]-Qo[:,4:]).max(),
          'quat dev', np.abs(Qg[:,:4]-Qo[:,:4]).max())

run('tiny', 10, 1.8, 1e30)
run('tiny', 10, 1.8, 9.0, outlier_frac=0.02)
run('tiny', 4, 4.0, 20.0, points_fixed=True, fixed=())
run('local', 10, 0.9, 1e30)
run('local', 6, 0.9, 7.25, outlier_frac=0.02)
