import sys, os, time, threading, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mageslam_amd import scene
from mageslam_amd.bundler import BundlerLib, load_scene
n = int(sys.argv[1]); steps = int(sys.argv[2]); tag = sys.argv[3] if len(sys.argv) > 3 else ""
bs = []
for i in range(n):
    s = scene.make_scene(n_cams=1000, n_pts=100000, n_obs=1000000, seed=0x5EED0004 + 0x100 * i)
    b = BundlerLib(False); load_scene(b, s, bulk=True); b.SetCurrentLambda(5e6)
    b.StepBundleAdjustment([1.8], 1e30, [])
    bs.append(b)
errs = []
def work(i):
    try:
        for k in range(steps):
            t0 = time.perf_counter(); bs[i].StepBundleAdjustment([1.8], 1e30, []); dt = time.perf_counter() - t0
            if dt > 0.2: print(tag, "slow step", i, k, round(dt, 3), flush=True)
    except Exception as e:
        errs.append((i, k, repr(e)))
th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
print(tag, json.dumps({"handles": n, "steps": steps, "seconds": round(time.perf_counter() - t0, 2), "errors": errs[:3]}), flush=True)
