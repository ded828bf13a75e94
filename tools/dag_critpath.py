#!/usr/bin/env python3
"""Critical path of a task-graph Cholesky trace (the input of tools/dag_trace.py): walks back from the last diagonal tile through, at every
step, the event that released it LAST -- a dependency (which one, and how long its hand-off took), or the team that took the task only then
(its previous task) -- and adds up where the launch's span went.    python tools/dag_critpath.py gpurun_out/dag_trace_6016.bin [-v]"""
import sys
import collections
import numpy as np

NAMES = {1: "strips", 2: "half", 3: "quarter", 4: "diag", 5: "ysolve", 6: "rhs"}


def tri(i, j):
    return i * (i + 1) // 2 + j


def main(path, verbose=False):
    raw = np.fromfile(path, dtype=np.int64)
    n_tasks, nt, qf, n_st = [int(v) for v in raw[:4]]
    tasks = raw[4:4 + n_tasks].view(np.uint64)
    st = raw[4 + n_tasks:4 + n_tasks + n_st].astype(float) * 0.01
    T = st[:4 * n_tasks].reshape(n_tasks, 4).copy()
    ch = st[4 * n_tasks:4 * n_tasks + 2 * nt].reshape(nt, 2).copy()
    t0 = ch[0, 0]
    T -= t0
    ch -= t0
    f = lambda s: ((tasks >> np.uint64(s)) & np.uint64(0xff)).astype(int)
    typ, I, J, U, K0, NK = f(0), f(8), f(16), f(24), f(32), f(40)
    # list boundaries: a new list starts where... the header does not carry them; rebuild from the library if it is there, else guess none
    glen = None
    try:
        import ctypes as C
        sys.path.insert(0, ".")
        from mageslam_amd import _lib
        L = _lib.lib()
        fn = L.mage_debug_chol_schedule
        fn.restype = C.c_int
        fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        out = np.zeros(n_tasks + 16, dtype=np.uint64)
        q = C.c_int(0)
        n_lists = int(getattr(L, "mage_debug_chol_n_lists", lambda: 9)())
        gl = (C.c_int * 32)()
        n = fn(nt, 256, 8, out.ctypes.data, out.size, C.byref(q), gl)
        if abs(n) == n_tasks and (out[:n_tasks] == tasks).all():
            glen = [gl[g] for g in range(n_lists)]
    except Exception as e:          # noqa: BLE001
        print("(no list boundaries:", e, ")")
    group = np.zeros(n_tasks, dtype=int)
    if glen:
        o = 0
        for g, n in enumerate(glen):
            group[o:o + n] = g
            o += n
    # ---- when every dependency word reached every value
    strip_end = collections.defaultdict(list)       # (i, k) -> [(end, id)]
    unit_steps = collections.defaultdict(list)      # (i, j, q) -> [(k_end, end, id)]
    diag_end = collections.defaultdict(list)
    ysol_end, yprog_steps = {}, collections.defaultdict(list)
    for t in range(n_tasks):
        ty, i, j, u, k0, nk = typ[t], I[t], J[t], U[t], K0[t], NK[t]
        e = T[t, 3]
        if ty == 1:
            strip_end[(i, j)].append((e, t))
        elif ty == 2:
            qs = [3] if (i == j and u == 1) else [2 * u, 2 * u + 1]
            for q in qs:
                unit_steps[(i, j, q)].append((k0 + nk, e, t))
        elif ty == 3:
            unit_steps[(i, j, u)].append((k0 + nk, e, t))
        elif ty == 4:
            diag_end[j].append((e, t))
        elif ty == 5:
            ysol_end[j] = (e, t)
        elif ty == 6:
            yprog_steps[i].append((k0 + nk, e, t))
    for v in unit_steps.values():
        v.sort()
    for v in yprog_steps.values():
        v.sort()

    def strips_done(i, k):          # (time, node)
        e, t = max(strip_end[(i, k)])
        return e, ("task", t)

    def unit_at(i, j, q, k):        # unit (i, j, q) stands at panel >= k
        if k <= 0:
            return 0.0, None
        for ke, e, t in unit_steps[(i, j, q)]:
            if ke >= k:
                return e, ("task", t)
        raise KeyError((i, j, q, k))

    def tile_at(i, j, k):           # every unit of the tile at panel >= k
        best = (0.0, None)
        for q in range(4):
            if i == j and q == 2:
                continue
            c = unit_at(i, j, q, k)
            if c[0] > best[0]:
                best = c
        return best

    def fact(k):
        return ch[k, 1], ("fact", k)

    def deps_of(t):
        ty, i, j, u, k0, nk = typ[t], I[t], J[t], U[t], K0[t], NK[t]
        d = []
        if ty == 1:
            d.append(("fact", fact(j)))
            if j > 0:
                d.append(("tile complete", tile_at(i, j, j)))
        elif ty == 5:
            d.append(("fact", fact(j)))
            if j > 0:
                for ke, e, tt in yprog_steps[j]:
                    if ke >= j:
                        d.append(("yprog", (e, ("task", tt))))
                        break
        elif ty == 6:
            if k0 > 0:
                for ke, e, tt in yprog_steps[i]:
                    if ke >= k0:
                        d.append(("yprog", (e, ("task", tt))))
                        break
            d.append(("strips(i)", strips_done(i, k0 + nk - 1)))
            e, tt = ysol_end[k0 + nk - 1]
            d.append(("ysol", (e, ("task", tt))))
        elif ty in (2, 3):
            qs = ([3] if (i == j and u == 1) else [2 * u, 2 * u + 1]) if ty == 2 else [u]
            for q in qs:
                d.append(("own unit", unit_at(i, j, q, k0)))
            d.append(("strips(i)", strips_done(i, k0 + nk - 1)))
            if i != j:
                d.append(("strips(j)", strips_done(j, k0 + nk - 1)))
        elif ty == 4:
            d.append(("strips(j,j-1)", strips_done(j, j - 1)))
            d.append(("diag tile", tile_at(j, j, j - 1)))
        return d

    # team identification: the task of the same list that ended last before this one was taken
    order = {}
    if glen:
        for g in range(len(glen)):
            ids = np.nonzero(group == g)[0]
            ends = T[ids, 3]
            srt = np.argsort(ends)
            order[g] = (ids[srt], ends[srt])

    def prev_on_team(t):
        if not glen:
            return None
        ids, ends = order[group[t]]
        k = np.searchsorted(ends, T[t, 0] + 0.02) - 1
        if k < 0:
            return None
        return int(ids[k])

    seg = collections.Counter()
    hops = collections.Counter()
    node = ("fact", nt - 1)
    now = ch[nt - 1, 1]
    steps = 0
    trail = []
    while node is not None and now > 0.5 and steps < 100000:
        steps += 1
        if node[0] == "fact":
            k = node[1]
            seg["potrf"] += ch[k, 1] - ch[k, 0]
            now = ch[k, 0]
            if k == 0:
                break
            e, t = max(diag_end[k])
            prev_store = ch[k - 1, 1]
            if e >= prev_store:
                seg["gather (9th arrival -> tile load starts)"] += now - e
                trail.append((now, "gather", k))
                node, now = ("task", t), e
            else:
                seg["chain: tile k-1 stored after its flag"] += now - prev_store
                node, now = ("fact", k - 1), prev_store
            continue
        t = node[1]
        ty = typ[t]
        name = NAMES[ty] + (" completing" if ty in (2, 3) and K0[t] + NK[t] == (J[t] - 1 if I[t] == J[t] else J[t]) else "") + (" near" if ty == 1 and K0[t] == 1 else "")
        seg["run " + name] += T[t, 3] - T[t, 2]
        seg["acquire"] += T[t, 2] - T[t, 1]
        now = T[t, 1]
        d = [(nm, v) for nm, v in deps_of(t) if v[1] is not None]
        best = max(d, key=lambda x: x[1][0]) if d else None
        if ty == 1 and K0[t] == 1:
            # phased near strips: they run behind the factorisation; what releases their END is fact(j)
            fe = ch[J[t], 1]
            if best is None or fe >= best[1][0] - 1e-9:
                # run segment double counts the part before fact: correct it
                seg["run " + name] -= max(0.0, min(fe, T[t, 3]) - T[t, 2])
                seg["acquire"] -= T[t, 2] - T[t, 1]
                hops["near strips <- fact"] += 1
                trail.append((T[t, 3], name, (I[t], J[t])))
                node, now = ("fact", J[t]), fe
                continue
        waited = T[t, 1] - T[t, 0]
        if best is not None and best[1][0] > T[t, 0] - 0.05:
            seg["hand-off (dep done -> consumer sees it): " + best[0]] += T[t, 1] - best[1][0]
            hops[name + " <- " + best[0]] += 1
            trail.append((T[t, 3], name, (I[t], J[t], U[t], K0[t], NK[t]), "dep " + best[0], round(T[t, 1] - best[1][0], 2)))
            node, now = best[1][1], best[1][0]
        else:
            seg["poll of ready deps"] += waited
            p = prev_on_team(t)
            if p is None:
                seg["start of launch"] += T[t, 0]
                break
            seg["take (prev task end -> this one taken)"] += T[t, 0] - T[p, 3]
            hops[name + " <- team busy"] += 1
            trail.append((T[t, 3], name, (I[t], J[t], U[t], K0[t], NK[t]), "team", round(T[t, 0] - T[p, 3], 2)))
            node, now = ("task", p), T[p, 3]
    total = sum(seg.values())
    print(f"critical path back from fact({nt - 1}) = {ch[nt - 1, 1]:.1f} us: {steps} steps, segments add up to {total:.1f} us (rest {now:.1f})")
    for k, v in sorted(seg.items(), key=lambda x: -x[1]):
        print(f"  {v:8.1f} us  {k}")
    print("edges on the path:")
    for k, v in sorted(hops.items(), key=lambda x: -x[1]):
        print(f"  {v:5d}  {k}")
    if verbose:
        for r in trail[::-1]:
            print(r)


if __name__ == "__main__":
    main(sys.argv[1], "-v" in sys.argv)
