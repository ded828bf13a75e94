"""ctypes binding of the window-sharded map driver (include/mage_window.h, mageslam_amd/csrc/window_host.hip): the C++ form of
mageslam_amd/windowed.py.  Used by the tests and tools/bench_windowed.py; the product is the shared library."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib
from .bundler import BundlerLib, _declare as _declare_ba

_f32 = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64 = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u32 = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u8 = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class _Params(C.Structure):
    _fields_ = [("n_windows", C.c_int), ("overlap", C.c_int), ("rank", C.c_int), ("world", C.c_int), ("device", C.c_int),
                ("threads", C.c_int)]


_declared = False


def _declare():
    global _declared
    if _declared:
        return
    _declare_ba()
    L = lib()
    vp, sz = C.c_void_p, C.c_size_t
    L.mage_wmap_create.argtypes = [C.POINTER(_Params), sz, _f32, _f32, _f32, _u8, sz, _f32, sz, _f32, _u32, _u32, _f32, C.POINTER(vp)]
    L.mage_wmap_destroy.argtypes = [vp]
    L.mage_wmap_destroy.restype = None
    L.mage_wmap_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp]
    L.mage_wmap_outer_iteration.argtypes = [vp, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_double)]
    L.mage_wmap_get_pose_block.argtypes = [vp, _f64]
    L.mage_wmap_pose_block_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.mage_wmap_window_info.argtypes = [vp, C.c_int, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(C.c_int)]
    L.mage_wmap_window_handle.argtypes = [vp, C.c_int, C.POINTER(vp)]
    _declared = True


class _BorrowedBundler(BundlerLib):
    """A window's mage_ba handle, owned by the map: diagnostics only (state, trace)."""

    def __init__(self, handle, n_cams, n_pts, n_obs):     # noqa: super().__init__ would create a handle
        _declare_ba()
        self._L = lib()
        self._h = C.c_void_p(handle)
        self.n_cams, self.n_pts, self.n_obs = n_cams, n_pts, n_obs

    def close(self):
        self._h = None


class WindowMap:
    def __init__(self, scene, n_windows: int, *, overlap: int = 0, rank: int = 0, world: int = 1, device: int = 0, threads: int = 1):
        _declare()
        self._L = lib()
        self._h = C.c_void_p()
        self.n_cams, self.n_windows = scene.n_cams, n_windows
        p = _Params(n_windows, overlap, rank, world, device, threads)
        check(self._L.mage_wmap_create(C.byref(p), scene.n_cams, np.ascontiguousarray(scene.cam_t, np.float32).reshape(-1),
                                       np.ascontiguousarray(scene.cam_R_colmajor(), np.float32).reshape(-1),
                                       np.ascontiguousarray(scene.cam_K, np.float32).reshape(-1),
                                       np.ascontiguousarray(scene.cam_fixed, np.uint8), scene.n_pts,
                                       np.ascontiguousarray(scene.points, np.float32).reshape(-1), scene.n_obs,
                                       np.ascontiguousarray(scene.obs_uv, np.float32).reshape(-1),
                                       np.ascontiguousarray(scene.obs_cam, np.uint32), np.ascontiguousarray(scene.obs_pt, np.uint32),
                                       np.ascontiguousarray(scene.obs_info, np.float32), C.byref(self._h)))
        self._cb = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.mage_wmap_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_allreduce(self, fn):
        """fn(block_ptr: int, count: int, stream: int) -> None, or None for a single rank."""
        if fn is None:
            self._cb = None
            check(self._L.mage_wmap_set_allreduce(self._h, C.cast(None, ALLREDUCE_FN), None))
            return

        def tramp(_ctx, ptr, count, stream):
            try:
                fn(int(ptr), int(count), int(stream or 0))
                return 0
            except Exception:                    # noqa: reported through the status code
                import traceback
                traceback.print_exc()
                return 1
        self._cb = ALLREDUCE_FN(tramp)
        check(self._L.mage_wmap_set_allreduce(self._h, self._cb, None))

    def outer_iteration(self, huber: float, max_err_sq: float = 1e30, inner: int = 1) -> float:
        v = C.c_double(0)
        check(self._L.mage_wmap_outer_iteration(self._h, float(huber), float(max_err_sq), int(inner), C.byref(v)))
        return float(v.value)

    def pose_block(self) -> np.ndarray:
        out = np.zeros((self.n_cams, 8))
        check(self._L.mage_wmap_get_pose_block(self._h, out.reshape(-1)))
        return out

    def window_info(self, w: int) -> dict:
        a, b, c, d = (C.c_size_t(0) for _ in range(4))
        o = C.c_int(0)
        check(self._L.mage_wmap_window_info(self._h, w, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(o)))
        return dict(own=a.value, cams=b.value, pts=c.value, obs=d.value, mine=bool(o.value))

    def window_bundler(self, w: int) -> BundlerLib:
        h = C.c_void_p()
        check(self._L.mage_wmap_window_handle(self._h, w, C.byref(h)))
        i = self.window_info(w)
        return _BorrowedBundler(h.value, i["cams"], i["pts"], i["obs"])
