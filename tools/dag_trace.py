#!/usr/bin/env python3
"""Reads gpurun_out/dag_trace_<n>.bin (tools/_bin/chol_test_trace, built with -DDAG_TRACE) and prints where the task-graph
Cholesky's time goes: per task type the count, the mean wait for dependencies, the acquire, the run time; the chain's period
per tile column; how many teams are busy over time.    python tools/dag_trace.py gpurun_out/dag_trace_6016.bin"""
import sys
import numpy as np

NAMES = {1: "strips", 2: "half", 3: "quarter", 4: "diag", 5: "ysolve", 6: "rhs"}


def main(path):
    raw = np.fromfile(path, dtype=np.int64)
    n_tasks, nt, qf, n_stamps = (int(v) for v in raw[:4])
    tasks = raw[4:4 + n_tasks].view(np.uint64)
    st = raw[4 + n_tasks:4 + n_tasks + n_stamps].astype(np.float64) * 0.01          # 100 MHz clock -> us
    T = st[:4 * n_tasks].reshape(n_tasks, 4)
    chain = st[4 * n_tasks:4 * n_tasks + 2 * nt].reshape(nt, 2)
    t0 = chain[0, 0]
    T = T - t0; chain = chain - t0
    typ = (tasks & np.uint64(0xff)).astype(int); nk = ((tasks >> np.uint64(40)) & np.uint64(0xff)).astype(int)
    jj = ((tasks >> np.uint64(16)) & np.uint64(0xff)).astype(int)
    end = max(T[:, 3].max(), chain[-1, 1])
    print(f"{path}: nt = {nt}, {n_tasks} tasks, quarter tiles from column {qf}; launch spans {end:.1f} us (chain ends {chain[-1, 1]:.1f}, last task {T[:, 3].max():.1f})")
    print("type      count   wait-deps   acquire   publish->run   run      run/panel   total-us(teams)")
    for t in sorted(NAMES):
        m = typ == t
        if not m.any():
            continue
        wait, acq, run = T[m, 1] - T[m, 0], T[m, 2] - T[m, 1], T[m, 3] - T[m, 2]
        per = run / np.maximum(nk[m], 1)
        print(f"{NAMES[t]:8s} {m.sum():6d}   {wait.mean():8.2f}   {acq.mean():7.2f}   {'':12s}   {run.mean():7.2f}   {per.mean():7.2f}   {(T[m, 3] - T[m, 0]).sum():10.0f}")
    for t in (1, 2, 3, 6):
        for n in sorted(set(nk[typ == t])):
            mm = (typ == t) & (nk == n)
            print(f"   {NAMES[t]} nk={n:2d}: {mm.sum():5d} tasks, run {np.mean(T[mm, 3] - T[mm, 2]):7.2f} us = {np.mean(T[mm, 3] - T[mm, 2]) / n:6.2f} per panel, wait {np.mean(T[mm, 1] - T[mm, 0]):6.2f}")
    if n_stamps >= 8 * n_tasks + 2 * nt + 16:
        T2 = st[4 * n_tasks + 2 * nt + 16:4 * n_tasks + 2 * nt + 16 + 4 * n_tasks].reshape(n_tasks, 4) - t0
        m = (typ == 1) & (T2[:, 2] > 0)
        if m.any():
            print("strips (late start, %d tasks): start->fact checked %.2f, park %.2f, compute %.2f, drain+sync %.2f" % (
                m.sum(), (T2[m, 0] - T[m, 2]).mean(), (T2[m, 1] - T2[m, 0]).mean(), (T2[m, 2] - T2[m, 1]).mean(), (T[m, 3] - T2[m, 2]).mean()))
    per = np.diff(chain[:, 1])
    print("chain: potrf (start->fact) mean %.2f us; period per column: first 10 %s ... mean %.2f, max %.2f" % ((chain[:, 1] - chain[:, 0]).mean(), np.round(per[:10], 1), per.mean(), per.max()))
    print("chain: wait before each tile (start_k - fact_{k-1}): mean %.2f  max %.2f" % ((chain[1:, 0] - chain[:-1, 1]).mean(), (chain[1:, 0] - chain[:-1, 1]).max()))
    # teams busy over time (pull -> end counts as busy; run only counts as running)
    W = 100.0
    nb = int(end // W) + 1
    busy, run = np.zeros(nb), np.zeros(nb)
    for a, b, arr in ((T[:, 0], T[:, 3], busy), (T[:, 2], T[:, 3], run)):
        for w in range(nb):
            lo, hi = w * W, (w + 1) * W
            arr[w] = np.clip(np.minimum(b, hi) - np.maximum(a, lo), 0, None).sum() / W
    col = [int((chain[:, 1] < (w + 1) * W).sum()) for w in range(nb)]
    print("window(us)  teams-holding-a-task  teams-running  chain-column")
    for w in range(nb):
        print(f"  {w * W:6.0f}      {busy[w]:7.1f}            {run[w]:7.1f}         {col[w]}")
    print("total team-us running %.0f = %.1f %% of 510 teams x span" % ((T[:, 3] - T[:, 2]).sum(), 100 * (T[:, 3] - T[:, 2]).sum() / (510 * end)))


if __name__ == "__main__":
    main(sys.argv[1])
