#!/usr/bin/env python3
"""Generates tests/golden/orb_bow.npz -- fixtures of the vocabulary tree's leaf lookup (OnlineBow::FindLeafNode, BoW/OnlineBow.cpp:289-311)
and of IndexedMatch through it (FeatureMatcher.cpp:192-292 with OnlineBow::QueryFeatures :115-132).  Development container only.

Descriptors are the committed ones of the 640x480 pair (orb_frames.npz).  Tree training (k-medoids, OnlineBow.cpp:325-500) is out of
scope, so the trees are toy vocabularies built by a fixed rule -- medoids = every k-th descriptor of image A in index order, assigned
level by level to their nearest parent -- with the awkward cases put in on purpose: a DUPLICATED medoid among siblings (two children
at the same distance: the first in the child list must win), a child list that is not in node order, a node with a single child, and a
one-level tree.  Leaf -> feature lists are filed the way the reference files an image (each descriptor under the leaf it descends to,
in index order).  Expected values come from the INDEPENDENT numpy restatement (oracle/indep/orb_numpy.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.indep import orb_numpy as N  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def toy_tree(desc, fanout, depth):
    """children_of (python lists) + node descriptors; node 0 is the root."""
    node_desc = [np.zeros(32, np.uint8)]
    children_of = [[]]
    frontier, pick = [0], 0
    for level in range(depth):
        nxt = []
        for parent in frontier:
            kids = []
            for _ in range(fanout if (parent % 5) else max(1, fanout - 2)):
                node_desc.append(desc[(7 * pick + 3 * level) % len(desc)].copy()); pick += 1
                children_of.append([])
                kids.append(len(node_desc) - 1)
            if len(kids) >= 3 and parent % 2 == 0:
                node_desc[kids[2]] = node_desc[kids[0]].copy()       # a tie between siblings: the first listed wins
                kids[0], kids[1] = kids[1], kids[0]                  # ... and the child list is not in node order
            children_of[parent] = kids
            nxt += kids
        frontier = nxt
    children_of[frontier[0]] = []
    return np.array(node_desc, np.uint8), children_of


def csr(ls):
    off = np.zeros(len(ls) + 1, np.int32)
    off[1:] = np.cumsum([len(l) for l in ls])
    return off, np.array([c for l in ls for c in l], np.int32)


def file_image(node_desc, children_of, desc):
    leaves = N.bow_find_leaf(node_desc, children_of, desc)
    per = [[] for _ in children_of]
    for i, l in enumerate(leaves):
        per[int(l)].append(i)
    return leaves, per


def main():
    g = np.load(os.path.join(HERE, "orb_frames.npz"))
    da, db = g["orb_640x480_a_desc"], g["orb_640x480_b_desc"]
    out = {}
    ma = (np.arange(len(da)) % 7) != 3
    mb = (np.arange(len(db)) % 11) != 5
    for name, fanout, depth in (("deep", 5, 3), ("flat", 9, 1), ("binary", 2, 5)):
        nd, ch = toy_tree(da, fanout, depth)
        la, fa = file_image(nd, ch, da)
        lb, fb = file_image(nd, ch, db)
        co, cc = csr(ch); fao, fac = csr(fa); fbo, fbc = csr(fb)
        out.update({f"{name}_nodes": nd, f"{name}_child_off": co, f"{name}_children": cc, f"{name}_leaf_a": la.astype(np.int32), f"{name}_leaf_b": lb.astype(np.int32),
                    f"{name}_feat_a_off": fao, f"{name}_feat_a": fac, f"{name}_feat_b_off": fbo, f"{name}_feat_b": fbc})
        for case, (md, mn, xa, xb) in {"plain": (30, 1, None, None), "loose": (64, 4, None, None), "masked": (50, 2, ma, mb)}.items():
            m = N.indexed_match_bow(nd, ch, da, fa, db, fb, md, mn, xa, xb)
            out[f"{name}_exp_{case}"] = m
            out[f"{name}_par_{case}"] = np.array([md, mn], np.int32)
            print(name, case, len(nd), "nodes", len(set(la.tolist())), "leaves hit by A", len(m), "matches")
    np.savez_compressed(os.path.join(HERE, "orb_bow.npz"), mask_a=ma, mask_b=mb, **out)


if __name__ == "__main__":
    main()
