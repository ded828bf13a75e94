"""Per-call latency of the host-pointer entry points a tracking thread would use: mage_orb_detect (one 640x480 frame in,
keypoints + descriptors out) and mage_match_bf (two descriptor sets in, matches out).

    python tools/orb_latency.py
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from mageslam_amd import frames
    from mageslam_amd.orb import Matcher, OrbDetector
    a, b = frames.make_frame(1, 640, 480), frames.make_frame(2, 640, 480)
    det, m = OrbDetector(), Matcher()
    for _ in range(20):
        ka, da = det.DetectAndCompute(a); kb, db = det.DetectAndCompute(b); m.Match(da, db, None, None, 30, 1)
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        det.DetectAndCompute(a)
    t1 = time.perf_counter()
    for _ in range(n):
        m.Match(da, db, None, None, 30, 1)
    t2 = time.perf_counter()
    print(f"mage_orb_detect 640x480 (host in, host out): {1e3*(t1-t0)/n:.3f} ms/frame; "
          f"mage_match_bf {len(da)}x{len(db)}: {1e3*(t2-t1)/n:.3f} ms/pair")


if __name__ == "__main__":
    main()
