import sys, os, json, hashlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from mageslam_amd import scene
from mageslam_amd.bundler import BundlerLib, load_scene
h = hashlib.sha256()
for cfg in ("local", "global"):
    s = scene.make_config(cfg)
    b = BundlerLib(False, device=0); load_scene(b, s, bulk=True)
    for _ in range(3): b.StepBundleAdjustment([1.8], 1e30, [])
    P = np.array([np.concatenate(b.GetPose(i)[0:1] + (b.GetPose(i)[1].ravel(),)) for i in range(0, s.n_cams, max(1, s.n_cams // 50))])
    h.update(P.tobytes()); h.update(np.float64([t["chi_after"] for t in b.trace()]).tobytes())
    b.close()
print("state+chi2 sha256", h.hexdigest()[:16], "gather" if os.environ.get("MAGE_BA_SCHUR_GATHER") else "staged",
      "materialised W" if os.environ.get("MAGE_BA_MATERIAL_W") else "compact W (the two fetch schemes only exist with MAGE_BA_MATERIAL_W=1)")
