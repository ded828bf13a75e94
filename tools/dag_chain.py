#!/usr/bin/env python3
"""Per tile column of a task-graph Cholesky trace (tools/dag_trace.py's input): when, relative to fact(k-1), the pieces on the chain's path
happened -- the strips of tile (k, k-1), the split panel, the start of potrf(k) -- and what they waited for.
    python tools/dag_chain.py gpurun_out/dag_trace_6016.bin [first_column last_column]"""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.int64)
n_tasks, nt, qf, n_st = [int(v) for v in raw[:4]]
tasks = raw[4:4 + n_tasks].view(np.uint64); st = raw[4 + n_tasks:4 + n_tasks + n_st].astype(float) * 0.01
T = st[:4 * n_tasks].reshape(n_tasks, 4); ch = st[4 * n_tasks:4 * n_tasks + 2 * nt].reshape(nt, 2)
t0 = ch[0, 0]; T -= t0; ch -= t0
typ = (tasks & np.uint64(0xff)).astype(int); I = ((tasks >> np.uint64(8)) & np.uint64(0xff)).astype(int); J = ((tasks >> np.uint64(16)) & np.uint64(0xff)).astype(int)
U = ((tasks >> np.uint64(24)) & np.uint64(0xff)).astype(int); K0 = ((tasks >> np.uint64(32)) & np.uint64(0xff)).astype(int); NK = ((tasks >> np.uint64(40)) & np.uint64(0xff)).astype(int)
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2, nt - 1)
print(" k period | tile(k,k-1) complete  strips: pulled ready end | diag tile complete  D: pulled ready end | potrf start   (all us after fact(k-1))")
for k in range(max(lo, 2), hi + 1):
    f = ch[k - 1, 1]
    ms = (typ == 1) & (I == k) & (J == k - 1); md = (typ == 4) & (J == k)
    mu = ((typ == 2) | (typ == 3)) & (I == k) & (J == k - 1) & (K0 + NK == k - 1)
    mu2 = ((typ == 2) | (typ == 3)) & (I == k) & (J == k) & (K0 + NK == k - 1)
    print("%2d %6.1f | %6.1f   %6.1f %6.1f %6.1f | %6.1f   %6.1f %6.1f %6.1f | %6.1f" % (
        k, ch[k, 1] - ch[k - 1, 1], T[mu, 3].max() - f, T[ms, 0].max() - f, T[ms, 1].max() - f, T[ms, 3].max() - f,
        (T[mu2, 3].max() - f) if mu2.any() else float("nan"), T[md, 0].max() - f, T[md, 1].max() - f, T[md, 3].max() - f, ch[k, 0] - f))
