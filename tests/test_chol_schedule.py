"""CPU tests of the task-graph Cholesky's HOST logic (mageslam_amd/csrc/chol_dag.hip, build_schedule / check_schedule): the static
task lists the persistent launch executes, one per group of teams.  No GPU: the library only has to load.  What the lists must satisfy
so that teams which take them in order cannot deadlock is re-derived here independently of the C++ checker."""
import ctypes as C

import numpy as np
import pytest

from mageslam_amd import _lib

T_STRIPS, T_HALF, T_QUARTER, T_DIAG, T_YSOLVE, T_RHS = 1, 2, 3, 4, 5, 6


def schedule(nt, n_cu=256, fuse=8):
    L = _lib.lib()
    f = L.mage_debug_chol_schedule
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    out = np.zeros(600000, dtype=np.uint64)
    qf = C.c_int(0)
    glen = (C.c_int * 9)()
    n = f(nt, n_cu, fuse, out.ctypes.data, out.size, C.byref(qf), glen)
    assert n > 0, "the library's own checker rejected its lists"
    w = out[:n].astype(np.uint64)
    cols = [(w >> np.uint64(s)) & np.uint64(0xff) for s in (0, 8, 16, 24, 32, 40)]
    flat = [tuple(int(c[i]) for c in cols) for i in range(n)]
    lists, o = [], 0
    for g in range(9):
        lists.append(flat[o:o + glen[g]]); o += glen[g]
    assert o == n
    return lists, qf.value


def units(i, j, qf):
    return 3 if i == j else 4          # 64 x 64 quarters; the diagonal tile has no upper-right one


def tile_group(i, j):
    return ((i >> 2) + 3 * (j >> 1)) & 7


@pytest.mark.parametrize("nt,n_cu,fuse", [(2, 256, 8), (3, 256, 8), (8, 256, 8), (13, 64, 4), (24, 256, 1), (47, 256, 8), (47, 304, 16), (60, 128, 8)])
def test_lists_are_complete_and_cannot_deadlock(nt, n_cu, fuse):
    lists, qf = schedule(nt, n_cu, fuse)
    tri = lambda i, j: i * (i + 1) // 2 + j
    stripc, usum, uprog, darr, yprog = {}, {}, {}, [0] * nt, [0] * nt
    state = {"fact": 1, "ysol": 0}
    cur = [0] * 9

    def try_run(g, task):
        typ, i, j, u, k0, nk = task
        if typ == T_STRIPS:
            assert i > j and nk == 4 and u in (0, 4) and k0 in (0, 1)
            if not (state["fact"] >= j + 1 and (j == 0 or usum.get(tri(i, j), 0) == units(i, j, qf) * j)):
                return False
            stripc[tri(i, j)] = stripc.get(tri(i, j), 0) + nk
        elif typ == T_YSOLVE:
            if not (state["fact"] >= j + 1 and yprog[j] == j):
                return False
            assert state["ysol"] == j
            state["ysol"] = j + 1
        elif typ == T_RHS:
            assert 1 <= nk <= fuse and k0 + nk <= i
            if not (yprog[i] >= k0 and stripc.get(tri(i, k0 + nk - 1), 0) == 8 and state["ysol"] >= k0 + nk):
                return False
            assert yprog[i] == k0
            yprog[i] = k0 + nk
        elif typ in (T_HALF, T_QUARTER):
            assert 1 <= nk <= fuse and i >= j >= 1 and k0 + nk <= (j - 1 if i == j else j)
            completing = k0 + nk == (j - 1 if i == j else j)
            assert tile_group(i, j) == g or (g == 8 and typ == T_QUARTER and i - j <= 2 and completing), "only completing quarters next to the diagonal are express tasks"
            if typ == T_HALF:          # column half u = quarters 2 u, 2 u + 1 (the diagonal tile's right half is quarter 3 alone)
                assert j < qf and u in (0, 1)
                qs = [3] if (i == j and u == 1) else [2 * u, 2 * u + 1]
                assert k0 + nk < (j - 1 if i == j else j), "the task that completes a tile goes in quarters"
            else:
                assert 0 <= u <= 3 and not (i == j and u == 2), "the diagonal tile has no upper-right quarter"
                qs = [u]
            if not (all(uprog.get((tri(i, j), q), 0) >= k0 for q in qs) and stripc.get(tri(i, k0 + nk - 1), 0) == 8 and stripc.get(tri(j, k0 + nk - 1), 0) == 8):
                return False
            for q in qs:
                assert uprog.get((tri(i, j), q), 0) == k0
                uprog[(tri(i, j), q)] = k0 + nk
            usum[tri(i, j)] = usum.get(tri(i, j), 0) + nk * len(qs)
        elif typ == T_DIAG:
            assert i == j and k0 == j - 1 and 0 <= u < 9
            if not (stripc.get(tri(j, j - 1), 0) == 8 and usum.get(tri(j, j), 0) == units(j, j, qf) * (j - 1)):
                return False
            darr[j] += 1
            while state["fact"] < nt and darr[state["fact"]] == 9:
                state["fact"] += 1
        else:
            raise AssertionError(f"unknown task type {typ}")
        return True

    moved = True
    while moved:
        moved = False
        for g in range(9):
            while cur[g] < len(lists[g]) and try_run(g, lists[g][cur[g]]):
                cur[g] += 1
                moved = True
    assert all(cur[g] == len(lists[g]) for g in range(9)), "a list's head waits for something no list will ever produce"
    assert state["fact"] == nt and state["ysol"] == nt and all(yprog[i] == i for i in range(nt))
    assert all(stripc.get(tri(i, j), 0) == 8 for i in range(1, nt) for j in range(i))
    assert all(usum.get(tri(i, j), 0) == units(i, j, qf) * (j - 1 if i == j else j) for i in range(1, nt) for j in range(1, i + 1))


def test_far_tiles_absorb_several_panels_per_task():
    """The point of the schedule: at the headline size the trailing update's tasks carry several panels where tiles lag behind the chain (a C block is
    read and written once per task, not once per panel), while the tiles next to the chain are served one panel at a time."""
    lists, qf = schedule(47, 256, 8)
    upd = [t for l in lists for t in l if t[0] in (T_HALF, T_QUARTER)]
    assert np.mean([t[5] for t in upd]) > 1.5 and max(t[5] for t in upd) == 8
    assert all(t[0] == T_QUARTER for t in upd if t[4] + t[5] == (t[2] - 1 if t[1] == t[2] else t[2])), "the task that completes a tile is a quarter"


def test_groups_carry_equal_shares():
    lists, qf = schedule(47, 256, 8)
    work = [sum(t[5] * (2 if t[0] == T_HALF else 1) for t in l if t[0] in (T_HALF, T_QUARTER)) for l in lists[:8]]      # in quarter-panels
    assert max(work) <= 1.08 * (sum(work) / 8)


def test_schedule_is_deterministic():
    assert schedule(24, 256, 8) == schedule(24, 256, 8)


def schedule_env(nt, env, n_cu=256, fuse=8):
    L = _lib.lib()
    f = L.mage_debug_chol_schedule_env
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    out = np.zeros(600000, dtype=np.uint64)
    qf = C.c_int(0)
    glen = (C.c_int * 9)()
    e = np.ascontiguousarray(env, dtype=np.int32)
    n = f(nt, n_cu, fuse, e.ctypes.data, out.ctypes.data, out.size, C.byref(qf), glen)
    w = out[:abs(n)].astype(np.uint64)
    cols = [(w >> np.uint64(s)) & np.uint64(0xff) for s in (0, 8, 16, 24, 32, 40)]
    return n, [tuple(int(c[i]) for c in cols) for i in range(abs(n))]


@pytest.mark.parametrize("nt,width", [(47, 2), (47, 0), (24, 5), (94, 3), (13, 12)])
def test_skyline_lists_touch_no_tile_left_of_the_envelope(nt, width):
    """mage_ba_use_skyline: the lists built from a skyline (env[i] = first tile column of tile row i that can hold a non-zero) pass the
    library's checker started from the envelope's initial progress, never name a tile or a panel left of the envelope, and still solve
    every strip, apply every panel and finish every rhs row inside it exactly once."""
    env = np.maximum(0, np.arange(nt) - width)
    n, tasks = schedule_env(nt, env)
    assert n > 0, "the library's own checker rejected the skyline lists"
    strips, panels = {}, {}
    for typ, i, j, u, k0, nk in tasks:
        if typ == T_STRIPS:
            assert j >= env[i], "strips of a tile left of the envelope"
            strips[(i, j)] = strips.get((i, j), 0) + nk
        elif typ in (T_HALF, T_QUARTER):
            assert j >= env[i] and k0 >= max(env[i], env[j]), "a panel that multiplies a structurally zero tile"
            for q in ([u] if typ == T_QUARTER else ([3] if (i == j and u == 1) else [2 * u, 2 * u + 1])):
                for k in range(k0, k0 + nk):
                    assert (i, j, q, k) not in panels
                    panels[(i, j, q, k)] = 1
        elif typ == T_RHS:
            assert k0 >= env[i]
    for i in range(1, nt):
        for j in range(env[i], i):
            assert strips.get((i, j), 0) == 8
        for j in range(max(env[i], 1), i + 1):
            for q in range(4):
                if i == j and q == 2:
                    continue
                for k in range(max(env[i], env[j]), (j - 1 if i == j else j)):
                    assert (i, j, q, k) in panels, (i, j, q, k)
    dense_n, _ = schedule_env(nt, np.zeros(nt, dtype=np.int32))
    assert width >= nt - 1 or n < dense_n


def test_random_skylines_are_accepted():
    rng = np.random.default_rng(11)
    for _ in range(25):
        nt = int(rng.integers(8, 70))
        env = np.array([int(rng.integers(0, i + 1)) for i in range(nt)])
        n, _ = schedule_env(nt, env, n_cu=int(rng.choice([64, 128, 256, 304])))
        assert n > 0

