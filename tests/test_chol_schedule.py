"""CPU tests of the task-graph Cholesky's HOST logic (mageslam_amd/csrc/chol_dag.hip, build_schedule / check_schedule): the static
task list the persistent launch executes.  No GPU: the library only has to load.  What a list must satisfy so that teams which take
tasks in list order cannot deadlock is re-derived here independently of the C++ checker."""
import ctypes as C

import numpy as np
import pytest

from mageslam_amd import _lib

T_STRIP, T_HALF, T_QUARTER, T_DIAG, T_YSOLVE = 1, 2, 3, 4, 5


def schedule(nt, n_cu=256, fuse=8):
    L = _lib.lib()
    f = L.mage_debug_chol_schedule
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    out = np.zeros(600000, dtype=np.uint64)
    qf = C.c_int(0)
    n = f(nt, n_cu, fuse, out.ctypes.data, out.size, C.byref(qf))
    assert n > 0, "the library's own checker rejected its list"
    w = out[:n].astype(np.uint64)
    cols = [(w >> np.uint64(s)) & np.uint64(0xff) for s in (0, 8, 16, 24, 32, 40)]
    return [tuple(int(c[i]) for c in cols) for i in range(n)], qf.value


def units(i, j, qf):
    return (3 if i == j else 4) if j >= qf else 2


@pytest.mark.parametrize("nt,n_cu,fuse", [(2, 256, 8), (3, 256, 8), (8, 256, 8), (13, 64, 4), (24, 256, 1), (47, 256, 8), (47, 304, 16), (60, 128, 8)])
def test_list_is_complete_and_topologically_ordered(nt, n_cu, fuse):
    tasks, qf = schedule(nt, n_cu, fuse)
    tri = lambda i, j: i * (i + 1) // 2 + j
    stripc, usum, uprog, darr = {}, {}, {}, [0] * nt
    fact, ysol = 1, 0
    for typ, i, j, u, k0, nk in tasks:
        if typ == T_STRIP:
            assert i > j and fact >= j + 1 and ysol >= j + 1
            assert j == 0 or usum.get(tri(i, j), 0) == units(i, j, qf) * j
            stripc[tri(i, j)] = stripc.get(tri(i, j), 0) + 1
        elif typ == T_YSOLVE:
            assert fact >= j + 1 and ysol == j and (j == 0 or stripc.get(tri(j, j - 1), 0) == 8)
            ysol = j + 1
        elif typ in (T_HALF, T_QUARTER):
            assert (typ == T_QUARTER) == (j >= qf) and 1 <= nk <= fuse and i >= j >= 1
            assert uprog.get((tri(i, j), u), 0) == k0 and k0 + nk <= (j - 1 if i == j else j)
            assert stripc.get(tri(i, k0 + nk - 1), 0) == 8 and stripc.get(tri(j, k0 + nk - 1), 0) == 8
            assert not (typ == T_QUARTER and i == j and u == 2), "the diagonal tile has no upper-right quarter"
            uprog[(tri(i, j), u)] = k0 + nk
            usum[tri(i, j)] = usum.get(tri(i, j), 0) + nk
        elif typ == T_DIAG:
            assert i == j and k0 == j - 1 and 0 <= u < 9 and stripc.get(tri(j, j - 1), 0) == 8
            assert usum.get(tri(j, j), 0) == units(j, j, qf) * (j - 1)
            darr[j] += 1
            while fact < nt and darr[fact] == 9:
                fact += 1
        else:
            raise AssertionError(f"unknown task type {typ}")
    assert fact == nt and ysol == nt
    assert all(stripc.get(tri(i, j), 0) == 8 for i in range(1, nt) for j in range(i))
    assert all(usum.get(tri(i, j), 0) == units(i, j, qf) * (j - 1 if i == j else j) for i in range(1, nt) for j in range(1, i + 1))


def test_far_tiles_absorb_several_panels_per_task():
    """The point of the schedule: at the headline size the trailing update's tasks carry more than two panels on average (a C block is
    read and written once per task, not once per panel), while the tiles next to the chain are served one panel at a time."""
    tasks, qf = schedule(47, 256, 8)
    upd = [t for t in tasks if t[0] in (T_HALF, T_QUARTER)]
    assert np.mean([t[5] for t in upd]) > 2.0
    assert all(t[5] == 1 for t in upd if t[2] == t[4] + 1), "column k + 1 takes panel k alone (it feeds the chain)"


def test_schedule_is_deterministic():
    assert schedule(24, 256, 8) == schedule(24, 256, 8)
