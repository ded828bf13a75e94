#!/usr/bin/env python3
"""Generates tests/golden/orb_indexed.npz -- IndexedMatch fixtures (development container only).

Descriptors are the committed ones of the 640x480 pair (orb_frames.npz); the candidate lists play the vocabulary index the
reference queries (BoW QueryFeatures, out of scope): word = top three bits of descriptor byte 0, a list holds the other image's
descriptors of the same word.  Every 5th list is reversed and every 9th repeats its first entry, because the outcome depends on
list order and on duplicates (strict '<' updates, FeatureMatcher.cpp:28-54).  Expected matches come from the INDEPENDENT numpy
restatement (oracle/indep/orb_numpy.py), so the fixture pins both the C oracle and the HIP kernel.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.indep import orb_numpy as N  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def lists(da, db):
    wa, wb = da[:, 0] >> 5, db[:, 0] >> 5
    out = []
    for i in range(len(da)):
        l = [int(v) for v in np.nonzero(wb == wa[i])[0]]
        if i % 5 == 0:
            l.reverse()
        if i % 9 == 0 and l:
            l.append(l[0])
        out.append(l)
    return out


def csr(ls):
    off = np.zeros(len(ls) + 1, np.int32)
    off[1:] = np.cumsum([len(l) for l in ls])
    return off, np.array([c for l in ls for c in l], np.int32)


def main():
    g = np.load(os.path.join(HERE, "orb_frames.npz"))
    da, db = g["orb_640x480_a_desc"], g["orb_640x480_b_desc"]
    cb, ca = lists(da, db), lists(db, da)
    ma = (np.arange(len(da)) % 7) != 3
    mb = (np.arange(len(db)) % 11) != 5
    cases = {"plain": (30, 1, None, None), "loose": (64, 4, None, None), "masked": (50, 2, ma, mb), "nodiff": (40, 0, None, mb)}
    out = {}
    for name, (md, mn, xa, xb) in cases.items():
        m = N.indexed_match(da, cb, db, ca, md, mn, xa, xb)
        out["exp_" + name] = m
        out["par_" + name] = np.array([md, mn], np.int32)
        print(name, len(m))
    bo, bc = csr(cb); ao, ac = csr(ca)
    np.savez_compressed(os.path.join(HERE, "orb_indexed.npz"), cand_b_off=bo, cand_b=bc, cand_a_off=ao, cand_a=ac, mask_a=ma, mask_b=mb, **out)


if __name__ == "__main__":
    main()
