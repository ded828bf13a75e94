#!/usr/bin/env python3
"""Development probe: where the cycles of k_fast_keypoints / k_select go.

Builds a SECOND copy of the library with -DMAGE_ORB_CLOCKS (mageslam_amd/_probe/libmageslam_hip_clk.so: thread 0 of every
workgroup adds the shader-clock time of each phase to a device array), runs the ORB detector of config 2 on it at batch 1 and
batch 256 frames and prints the phase split.  The product library is not touched.

    python tools/orb_phase_probe.py            (on the GPU box)
    python tools/orb_phase_probe.py --build    (here: compile only)
"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBE_DIR = os.path.join(ROOT, "mageslam_amd", "_probe")
PROBE_LIB = os.path.join(PROBE_DIR, "libmageslam_hip_clk.so")


def build():
    from mageslam_amd import build as B
    os.makedirs(PROBE_DIR, exist_ok=True)
    objs = []
    for src in B.sources():
        obj = os.path.join(PROBE_DIR, os.path.basename(src) + ".o")
        if not os.path.exists(obj) or os.path.getmtime(obj) < B._deps_mtime():
            subprocess.check_call([B.HIPCC, *B.flags_for(src), "-DMAGE_ORB_CLOCKS", "-c", src, "-o", obj])
        objs.append(obj)
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROBE_LIB, *objs])


FAST = {0: "zero + stage window", 1: "compass test + compaction", 2: "pair test + exact score of survivors", 3: "3x3 NMS + horizontal blur pass + keypoint output", 4: "vertical blur pass"}
SELECT = {15: "zero + tile-count scan", 16: "entries into registers", 17: "histogram", 18: "suffix scan", 10: "thresholds", 19: "cell counts", 20: "cell scan", 12: "cell scatter", 11: "compaction of candidates + bounding box", 13: "ring search (radii)", 14: "rank + output"}


def main():
    if "--build" in sys.argv:
        build(); print(PROBE_LIB); return
    import numpy as np
    import torch
    from mageslam_amd import _lib
    _lib.LIB_PATH = PROBE_LIB
    from mageslam_amd import frames
    from mageslam_amd.orb import OrbDetector
    L = _lib.lib()
    base = [frames.frame_pair(500 + i) for i in range(8)]
    allf = np.concatenate([np.stack([p[0] for p in base]), np.stack([p[1] for p in base])])
    det = OrbDetector()
    det.enable_profile(True)
    out = {}
    for nf in (1, 512):
        imgs = torch.from_numpy(allf[np.arange(nf) % 16]).cuda().contiguous()
        for _ in range(2):
            det.detect_batch_device(imgs.data_ptr(), nf, 640, 480, 440)
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 32)()
        assert L.mage_orb_debug_clocks(buf, 1) == 0
        reps = 10
        for _ in range(reps):
            det.detect_batch_device(imgs.data_ptr(), nf, 640, 480, 440)
        torch.cuda.synchronize()
        assert L.mage_orb_debug_clocks(buf, 1) == 0
        p = det.profile()
        for name, tab in (("k_fast_keypoints", FAST), ("k_select", SELECT)):
            tot = sum(buf[i] for i in tab) or 1
            out[f"{name} frames={nf}"] = {v: round(buf[i] / tot, 3) for i, v in tab.items()}
            out[f"{name} frames={nf}"]["ticks_all_sampled_workgroups"] = tot
        out[f"sampled workgroups frames={nf}"] = buf[31]
        out[f"stage_ms frames={nf}"] = {"fast": p.fast_ms, "nms_select": p.select_ms, "blur": p.blur_ms, "brief": p.brief_ms}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
