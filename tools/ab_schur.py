#!/usr/bin/env python3
"""A/B of the Schur build's two launches on the 1k-pose map: one stream of trips per resident wavefront (k_schur_stream, the default) against
one wavefront per block (MAGE_BA_SCHUR_BLOCKS=1): the stage's HIP-event span over 20 LM iterations and a hash of the final state (the
two must agree to the bit).    python tools/ab_schur.py [workload]"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(workload):
    sys.path.insert(0, ROOT)
    from mageslam_amd import scene
    from mageslam_amd.bundler import BundlerLib, load_scene
    s = scene.make_config(workload)
    b = BundlerLib(False, device=0)
    load_scene(b, s, bulk=True)
    b.enable_profiling(True)
    out = []
    for _ in range(20):
        b.StepBundleAdjustment([1.8], 1e30, out)
    p = b.profile()
    h = hashlib.sha256(b.poses_f64().tobytes() + b.points_f64().tobytes()).hexdigest()[:16]
    print(json.dumps({"schur_ms": p.schur_ms_total / max(p.schur_launches, 1), "launches": p.schur_launches, "state_hash": h,
                      "factor_ms": p.factor_ms_total / max(p.n_factorizations, 1)}))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--child":
        child(sys.argv[1])
        sys.exit(0)
    wl = sys.argv[1] if len(sys.argv) > 1 else "global"
    res = {}
    for name, env in (("stream", {}), ("per_block", {"MAGE_BA_SCHUR_BLOCKS": "1"})):
        e = dict(os.environ, **env)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), wl, "--child"], capture_output=True, text=True, env=e)
        if p.returncode != 0:
            print(name, "FAILED", p.stderr[-1500:])
            sys.exit(1)
        res[name] = json.loads(p.stdout.strip().splitlines()[-1])
    res["bit_identical"] = res["stream"]["state_hash"] == res["per_block"]["state_hash"]
    print(json.dumps(res))
    sys.exit(0 if res["bit_identical"] else 1)
