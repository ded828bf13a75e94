#!/usr/bin/env python3
"""Development probe: what each phase of k_fast_keypoints costs IN PLACE.

Builds copies of the library whose orb_kernels.hip is compiled with -DMAGE_ORB_ABLATE=<bits> (a phase's arithmetic run twice, or
a phase left out; see the MAGE_ORB_ABLATE blocks in orb_kernels.hip -- the results of such a build are wrong on purpose) and times
the FAST launch of a 2048-frame batch on each.  The product library is not touched.

    python tools/orb_ablate.py --build 0 1 2 4 8     (here: compile the variants)
    python tools/orb_ablate.py 0 1 2 4 8             (on the GPU box: time them; one process per variant)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBE_DIR = os.path.join(ROOT, "mageslam_amd", "_probe")


def lib_path(bits):
    return os.path.join(PROBE_DIR, f"libmageslam_hip_ablate{bits}.so")


def build(bits):
    from mageslam_amd import build as B
    B.build()                                             # the objects of everything else
    os.makedirs(PROBE_DIR, exist_ok=True)
    objs = []
    for src in B.sources():
        if os.path.basename(src) == "orb_kernels.hip":
            obj = os.path.join(PROBE_DIR, f"orb_kernels.ablate{bits}.o")
            subprocess.check_call([B.HIPCC, *B.flags_for(src), f"-DMAGE_ORB_ABLATE={bits}", "-c", src, "-o", obj])
        else:
            obj = os.path.join(B.OBJ, os.path.basename(src) + ".o")
        objs.append(obj)
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path(bits), *objs])


def run_one(bits):
    import numpy as np
    import torch
    from mageslam_amd import _lib
    _lib.LIB_PATH = bits if isinstance(bits, str) else lib_path(bits)
    from mageslam_amd import frames
    from mageslam_amd.orb import OrbDetector
    base = [frames.frame_pair(500 + i) for i in range(8)]
    allf = np.concatenate([np.stack([p[0] for p in base]), np.stack([p[1] for p in base])])
    det = OrbDetector()
    det.enable_profile(True)
    nf = 2048
    imgs = torch.from_numpy(allf[np.arange(nf) % 16]).cuda().contiguous()
    ms, sel, brief = [], [], []
    for i in range(int(os.environ.get("ORB_ABLATE_REPS", "8"))):
        det.detect_batch_device(imgs.data_ptr(), nf, 640, 480, 440)
        torch.cuda.synchronize()
        if i >= 2:
            p = det.profile()
            ms.append(p.fast_ms); sel.append(p.select_ms); brief.append(p.brief_ms)
    one = []
    img1 = imgs[:1].contiguous()
    for i in range(60):
        det.detect_batch_device(img1.data_ptr(), 1, 640, 480, 440)
        torch.cuda.synchronize()
        if i >= 10:
            one.append(det.profile().select_ms)
    print(json.dumps({"ablate": bits, "fast_ms_median": float(np.median(ms)), "fast_ms_min": float(min(ms)), "select_ms_median": float(np.median(sel)),
                      "brief_ms_median": float(np.median(brief)), "select_one_frame_us_median": 1e3 * float(np.median(one))}))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--build" in sys.argv:
        for b in args:
            build(int(b)); print(lib_path(int(b)))
        return
    if "--one" in sys.argv:
        run_one(int(args[0]) if args[0].isdigit() else args[0]); return   # a number: that variant; otherwise the path of a library
    for b in args:
        subprocess.call([sys.executable, os.path.abspath(__file__), "--one", b])


if __name__ == "__main__":
    main()
