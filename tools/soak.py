"""Soak test: many create / solve / destroy cycles of mixed problem sizes on concurrent threads; checks that results repeat
bit for bit and that parked memory stays bounded.    python tools/soak.py [--cycles 60]"""
import argparse, hashlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mageslam_amd import scene
from mageslam_amd.bundler import BundlerLib, load_scene, release_cached_memory


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--cycles", type=int, default=60); a = ap.parse_args()
    shapes = [dict(n_cams=10, n_pts=200, n_obs=2000, seed=1), dict(n_cams=20, n_pts=5000, n_obs=50000, seed=3),
              dict(n_cams=150, n_pts=15000, n_obs=150000, seed=5), dict(n_cams=400, n_pts=40000, n_obs=400000, seed=7)]
    scenes = [scene.make_scene(**k) for k in shapes]
    ref = {}
    errors = []

    def run(i):
        s = scenes[i]
        b = BundlerLib(False); load_scene(b, s, bulk=True)
        out = []
        for _ in range(3):
            b.StepBundleAdjustment([1.8], 1e30 if i else 25.0, out)
        h = hashlib.sha256(b.poses_f64().tobytes() + b.points_f64().tobytes() + np.array(out, np.uint32).tobytes()).hexdigest()
        return h

    for i in range(len(scenes)):
        ref[i] = run(i)

    def worker(tid):
        rng = np.random.default_rng(tid)
        for c in range(a.cycles):
            i = int(rng.integers(0, len(scenes)))
            try:
                h = run(i)
                if h != ref[i]:
                    errors.append((tid, c, i, "result changed"))
            except BaseException as e:   # noqa: BLE001
                errors.append((tid, c, i, repr(e)))
            if tid == 0 and c % 20 == 19:
                release_cached_memory()

    t0 = time.time()
    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in th: t.start()
    for t in th: t.join()
    import subprocess
    mem = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
    print(f"{4 * a.cycles} solves on 4 threads in {time.time() - t0:.1f} s, errors: {errors[:3] if errors else 'none'}; vram: {mem}")
    sys.exit(1 if errors else 0)


if __name__ == "__main__":
    main()
