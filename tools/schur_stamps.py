#!/usr/bin/env python3
"""Per-wavefront stamps of k_schur_stream on the 1k-pose map (MAGE_BA_SCHUR_TRACE=1: start, end on the 100 MHz clock, hardware id, trips,
blocks, diagonal trips per wavefront; eight wavefronts per workgroup = compute unit): where the wavefronts sit, when they end, how far
apart the two wavefronts of a SIMD and the compute units end, and a least-squares fit of a unit's end time to its trips / blocks.
    MAGE_BA_SCHUR_TRACE=1 python tools/schur_stamps.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mageslam_amd import scene  # noqa: E402
from mageslam_amd.bundler import BundlerLib, load_scene  # noqa: E402


def main():
    if not os.environ.get("MAGE_BA_SCHUR_TRACE"):
        sys.exit("set MAGE_BA_SCHUR_TRACE=1")
    s = scene.make_config("global")
    b = BundlerLib(False, device=0)
    load_scene(b, s, bulk=True)
    b.SetCurrentLambda(5e6)
    out = []
    for _ in range(3):
        b.StepBundleAdjustment([1.8], 1e30, out)
    st = b.structure("stream_stamps").view(np.int64).reshape(-1, 4)
    t0 = st[:, 0].min()
    start, end = (st[:, 0] - t0) * 0.01, (st[:, 1] - t0) * 0.01
    trips, nblk, ndiag = st[:, 3] & 0xffffffff, (st[:, 3] >> 32) & 0xffff, st[:, 3] >> 48
    hw, xcc = st[:, 2] & 0xffffffff, st[:, 2] >> 32
    dur = end - start
    print(f"{len(st)} wavefronts; last start {start.max():.1f} us; end min {end.min():.1f} mean {end.mean():.1f} max {end.max():.1f} us")
    print(f"trips per wavefront mean {trips.mean():.1f} (min {trips.min()}, max {trips.max()}), blocks mean {nblk.mean():.1f}; {dur.sum() / trips.sum():.3f} us per trip per wavefront")
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7          # HW_ID: wave_id[3:0] simd_id[5:4] cu_id[11:8] sh_id[12] se_id[15:13]
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    per_cu = np.bincount(np.unique(key, return_inverse=True)[1])
    per_simd = np.bincount(np.unique(key * 4 + simd, return_inverse=True)[1])
    g = np.arange(len(st)) // 8
    print(f"{len(per_cu)} compute units hold {per_cu.min()}..{per_cu.max()} wavefronts, {len(per_simd)} SIMDs {per_simd.min()}..{per_simd.max()}; XCC id == workgroup % 8 for {(xcc == g % 8).mean():.3f} of them")
    pair = {}
    for w in range(len(st)):
        pair.setdefault(int(key[w] * 4 + simd[w]), []).append(w)
    pd = np.array([[end[v[0]], end[v[1]]] for v in pair.values() if len(v) == 2])
    print(f"the two wavefronts of a SIMD end {np.abs(pd[:, 0] - pd[:, 1]).mean():.1f} us apart on average (max {np.abs(pd[:, 0] - pd[:, 1]).max():.1f})")
    ng = int(g.max()) + 1
    T, B, D = np.bincount(g, trips, ng), np.bincount(g, nblk, ng), np.bincount(g, ndiag, ng)
    last = np.array([end[g == k].max() for k in range(ng)])
    print(f"compute units end {last.min():.1f} .. {last.max():.1f} us (mean {last.mean():.1f}); per XCD mean {np.round([last[x::8].mean() for x in range(8)], 1)}")
    print(f"per unit: trips {int(T.min())}..{int(T.max())}, blocks {int(B.min())}..{int(B.max())}, diagonal trips {int(D.min())}..{int(D.max())}")
    A = np.stack([T, B, D, np.ones(ng)], 1)
    coef, *_ = np.linalg.lstsq(A, last, rcond=None)
    print("fit: unit end = %.3f us x trips + %.3f x blocks + %.3f x diagonal trips + %.1f (rms %.2f us)" % (*coef, np.sqrt(np.mean((A @ coef - last) ** 2))))


if __name__ == "__main__":
    main()
