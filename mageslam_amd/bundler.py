"""Python mirror of the reference's ``mage::BundlerLib`` facade (Dependencies/BundlerLib/Include/BundlerLib.h:20-66)
over the C ABI of libmageslam_hip.so (include/mage_ba.h).  Same method names, same argument meaning,
same call protocol; used by the tests and by bench.py.  The product itself is the shared library and
the C++ shim include/BundlerLib.h -- this file is only a binding.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import MAGE_OK, check, lib

_f32 = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64 = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u32 = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_u8 = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


class _Params(C.Structure):
    _fields_ = [("are_points_fixed", C.c_int), ("device", C.c_int)]


class IterStats(C.Structure):
    _fields_ = [("code", C.c_int), ("trials", C.c_int), ("chi2_before", C.c_double), ("chi2_after", C.c_double),
                ("lambda_", C.c_double)]


class Profile(C.Structure):
    _fields_ = [("n_factorizations", C.c_uint64), ("factor_ms_total", C.c_double), ("factor_flops_each", C.c_double),
                ("schur_launches", C.c_uint64), ("schur_ms_total", C.c_double), ("system_order", C.c_int),
                ("padded_order", C.c_int), ("linearize_launches", C.c_uint64), ("linearize_ms_total", C.c_double),
                ("linearize_bytes_each", C.c_double), ("schur_bytes_each", C.c_double), ("update_launches", C.c_uint64),
                ("update_ms_total", C.c_double), ("update_bytes_each", C.c_double),
                ("trials_rerun_after_stall", C.c_uint64), ("fallback_to_separate_launches", C.c_uint64)]


_declared = False


def _declare():
    global _declared
    if _declared:
        return
    L = lib()
    vp, sz = C.c_void_p, C.c_size_t
    L.mage_ba_create.argtypes = [C.POINTER(_Params), C.POINTER(vp)]
    L.mage_ba_destroy.argtypes = [vp]
    L.mage_ba_destroy.restype = None
    for n in ("cameras", "points", "observations", "fixed_distance_constraints", "relative_rotation_constraints",
              "relative_transform_constraints"):
        getattr(L, "mage_ba_alloc_" + n).argtypes = [vp, sz]
    L.mage_ba_set_camera.argtypes = [vp, sz, _f32, _f32, _f32, C.c_int]
    L.mage_ba_fix_camera.argtypes = [vp, sz, C.c_int]
    L.mage_ba_set_point.argtypes = [vp, sz, _f32]
    L.mage_ba_set_observation.argtypes = [vp, sz, _f32, C.c_uint64, C.c_uint64, C.c_float]
    L.mage_ba_set_cameras_bulk.argtypes = [vp, sz, _f32, _f32, _f32, _u8]
    L.mage_ba_set_points_bulk.argtypes = [vp, sz, _f32]
    L.mage_ba_update_camera_poses.argtypes = [vp, sz, _u32, _f32, _f32]
    L.mage_ba_set_observations_bulk.argtypes = [vp, sz, _f32, _u32, _u32, _f32]
    L.mage_ba_set_fixed_distance_constraint.argtypes = [vp, sz, sz, sz, C.c_float, C.c_float]
    L.mage_ba_set_relative_rotation_constraint.argtypes = [vp, sz, sz, sz, _f32, C.c_float]
    L.mage_ba_set_relative_transform_constraint.argtypes = [vp, sz, sz, sz, _f32, _f32, C.c_float]
    L.mage_ba_set_lambda.argtypes = [vp, C.c_float]
    L.mage_ba_get_lambda.argtypes = [vp, C.POINTER(C.c_float)]
    L.mage_ba_step.argtypes = [vp, _f32, sz, C.c_float, _u32, sz, C.POINTER(sz), C.POINTER(C.c_float)]
    L.mage_ba_get_outliers.argtypes = [vp, C.c_void_p, sz, C.POINTER(sz)]
    L.mage_ba_bind_pose_exchange.argtypes = [vp, sz, _u32, _u32, sz, _u32, _u32]
    L.mage_ba_export_poses_device.argtypes = [vp, vp, vp]
    L.mage_ba_import_poses_device.argtypes = [vp, vp, vp]
    L.mage_ba_synchronize.argtypes = [vp]
    L.mage_ba_get_pose.argtypes = [vp, sz, _f32, _f32]
    L.mage_ba_get_point.argtypes = [vp, sz, _f32]
    L.mage_ba_get_poses_bulk.argtypes = [vp, sz, _f32, _f32]
    L.mage_ba_get_points_bulk.argtypes = [vp, sz, _f32]
    L.mage_ba_get_state_f64.argtypes = [vp, _f64, _f64]
    L.mage_ba_get_iter_stats.argtypes = [vp, C.POINTER(IterStats), sz, C.POINTER(sz)]
    L.mage_ba_enable_profiling.argtypes = [vp, C.c_int]
    L.mage_ba_use_skyline.argtypes = [vp, C.c_int]
    L.mage_ba_debug_schur_per_block.argtypes = [vp, C.c_int]
    L.mage_ba_get_profile.argtypes = [vp, C.POINTER(Profile)]
    L.mage_ba_debug_structure.argtypes = [vp, C.c_char_p, C.c_void_p, sz, C.POINTER(sz)]
    L.mage_release_cached_memory.argtypes = []
    L.mage_release_cached_memory.restype = None
    _declared = True


def release_cached_memory() -> None:
    """Returns the device / pinned buffers and streams parked by destroyed handles to the HIP runtime (include/mage_ba.h)."""
    _declare()
    lib().mage_release_cached_memory()


class BundlerLib:
    """``mage::BundlerLib`` on one MI355X.  ``device`` is the HIP ordinal (-1: current device)."""

    def __init__(self, are_points_fixed: bool = False, device: int = -1):
        _declare()
        self._L = lib()
        self._h = C.c_void_p()
        p = _Params(int(are_points_fixed), int(device))
        check(self._L.mage_ba_create(C.byref(p), C.byref(self._h)))
        self.n_cams = self.n_pts = self.n_obs = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.mage_ba_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    # --- BundlerLib.h:25-47
    def AllocateCameras(self, count): check(self._L.mage_ba_alloc_cameras(self._h, count)); self.n_cams = count

    def SetCameraPose(self, idx, position, orientation_colmajor, intrinsics, is_fixed):
        check(self._L.mage_ba_set_camera(self._h, idx, np.ascontiguousarray(position, np.float32),
                                         np.ascontiguousarray(orientation_colmajor, np.float32).reshape(9),
                                         np.ascontiguousarray(intrinsics, np.float32), int(is_fixed)))

    def FixCameraPose(self, idx, value): check(self._L.mage_ba_fix_camera(self._h, idx, int(value)))

    def UpdateCameraPoses(self, indices, positions, orientations_colmajor):
        """Extension (include/mage_ba.h): pose-only re-seed of cameras already in the graph; no structure rebuild."""
        idx = np.ascontiguousarray(indices, np.uint32)
        check(self._L.mage_ba_update_camera_poses(self._h, len(idx), idx, np.ascontiguousarray(positions, np.float32).reshape(-1),
                                                  np.ascontiguousarray(orientations_colmajor, np.float32).reshape(-1)))

    def AllocateMapPoints(self, count): check(self._L.mage_ba_alloc_points(self._h, count)); self.n_pts = count
    def SetMapPoint(self, idx, point): check(self._L.mage_ba_set_point(self._h, idx, np.ascontiguousarray(point, np.float32)))
    def AllocateObservations(self, count): check(self._L.mage_ba_alloc_observations(self._h, count)); self.n_obs = count

    def SetObservation(self, idx, position, camera_index, map_point_index, information_scalar):
        check(self._L.mage_ba_set_observation(self._h, idx, np.ascontiguousarray(position, np.float32),
                                              int(camera_index), int(map_point_index), float(information_scalar)))

    def AllocateFixedDistanceConstraints(self, count): check(self._L.mage_ba_alloc_fixed_distance_constraints(self._h, count))
    def AllocateRelativeRotationConstraints(self, count): check(self._L.mage_ba_alloc_relative_rotation_constraints(self._h, count))
    def AllocateRelativeTransformConstraints(self, count): check(self._L.mage_ba_alloc_relative_transform_constraints(self._h, count))

    # quaternions: Eigen::Quaternionf coefficient order x, y, z, w (BundlerLib.h:40-47)
    def SetFixedDistanceConstraint(self, idx, camera_index_1, camera_index_2, distance=1.0, weight=1.0):
        check(self._L.mage_ba_set_fixed_distance_constraint(self._h, idx, int(camera_index_1), int(camera_index_2), float(distance), float(weight)))

    def SetRelativeRotationConstraint(self, idx, camera_index_1, camera_index_2, delta_rotation_xyzw, weight=1.0):
        check(self._L.mage_ba_set_relative_rotation_constraint(self._h, idx, int(camera_index_1), int(camera_index_2),
                                                               np.ascontiguousarray(delta_rotation_xyzw, np.float32), float(weight)))

    def SetRelativeTransformConstraint(self, idx, camera_index_1, camera_index_2, delta_position, delta_rotation_xyzw, weight):
        check(self._L.mage_ba_set_relative_transform_constraint(self._h, idx, int(camera_index_1), int(camera_index_2),
                                                                np.ascontiguousarray(delta_position, np.float32),
                                                                np.ascontiguousarray(delta_rotation_xyzw, np.float32), float(weight)))

    # --- bulk setters (mage_ba_set_*_bulk)
    def SetCamerasBulk(self, positions, R_colmajor, intrinsics, fixed):
        n = len(positions)
        check(self._L.mage_ba_set_cameras_bulk(self._h, n, np.ascontiguousarray(positions, np.float32).reshape(-1),
                                               np.ascontiguousarray(R_colmajor, np.float32).reshape(-1),
                                               np.ascontiguousarray(intrinsics, np.float32).reshape(-1),
                                               np.ascontiguousarray(fixed, np.uint8)))

    def SetMapPointsBulk(self, points):
        check(self._L.mage_ba_set_points_bulk(self._h, len(points), np.ascontiguousarray(points, np.float32).reshape(-1)))

    def SetObservationsBulk(self, uv, cam, pt, info):
        check(self._L.mage_ba_set_observations_bulk(self._h, len(uv), np.ascontiguousarray(uv, np.float32).reshape(-1),
                                                    np.ascontiguousarray(cam, np.uint32), np.ascontiguousarray(pt, np.uint32),
                                                    np.ascontiguousarray(info, np.float32)))

    # --- BundlerLib.h:49-59
    def SetCurrentLambda(self, user_lambda): check(self._L.mage_ba_set_lambda(self._h, float(user_lambda)))

    def GetCurrentLambda(self) -> float:
        v = C.c_float()
        check(self._L.mage_ba_get_lambda(self._h, C.byref(v)))
        return float(v.value)

    def StepBundleAdjustment(self, huber_width_per_iteration, max_error_square, outliers: list) -> float:
        hw = np.ascontiguousarray(huber_width_per_iteration, np.float32)
        buf = getattr(self, "_out_buf", None)
        if buf is None or buf.size < max(self.n_obs, 1):          # one buffer per handle, sized like the reference's ReserveOutliers
            buf = self._out_buf = np.zeros(max(self.n_obs, 1), np.uint32)
        n, mse = C.c_size_t(0), C.c_float(0)
        check(self._L.mage_ba_step(self._h, hw, hw.size, float(max_error_square), buf, buf.size, C.byref(n), C.byref(mse)))
        if n.value:
            outliers.extend(buf[: min(n.value, buf.size)].tolist())
        return float(mse.value)

    def GetOutliers(self) -> list:
        """mage_ba_get_outliers: the complete outlier list of the most recent step."""
        n = C.c_size_t(0)
        check(self._L.mage_ba_get_outliers(self._h, None, 0, C.byref(n)))
        buf = np.zeros(max(n.value, 1), np.uint32)
        check(self._L.mage_ba_get_outliers(self._h, buf.ctypes.data, buf.size, C.byref(n)))
        return [int(x) for x in buf[: n.value]]

    # --- device-resident pose exchange (mage_ba.h): block_ptr / stream are raw device pointer / hipStream_t values
    def BindPoseExchange(self, export_cameras, export_rows, import_cameras, import_rows):
        ec, er = np.ascontiguousarray(export_cameras, np.uint32), np.ascontiguousarray(export_rows, np.uint32)
        ic, ir = np.ascontiguousarray(import_cameras, np.uint32), np.ascontiguousarray(import_rows, np.uint32)
        check(self._L.mage_ba_bind_pose_exchange(self._h, len(ec), ec, er, len(ic), ic, ir))

    def ExportPosesDevice(self, block_ptr: int, consumer_stream: int = 0):
        check(self._L.mage_ba_export_poses_device(self._h, C.c_void_p(block_ptr), C.c_void_p(consumer_stream or None)))

    def ImportPosesDevice(self, block_ptr: int, producer_stream: int = 0):
        check(self._L.mage_ba_import_poses_device(self._h, C.c_void_p(block_ptr), C.c_void_p(producer_stream or None)))

    def Synchronize(self): check(self._L.mage_ba_synchronize(self._h))

    def GetPose(self, idx):
        t = np.zeros(3, np.float32); R = np.zeros(9, np.float32)
        check(self._L.mage_ba_get_pose(self._h, idx, t, R))
        return t, R

    def GetPoint(self, idx):
        p = np.zeros(3, np.float32)
        check(self._L.mage_ba_get_point(self._h, idx, p))
        return p

    def GetPosesBulk(self):
        t = np.zeros((self.n_cams, 3), np.float32); R = np.zeros((self.n_cams, 9), np.float32)
        check(self._L.mage_ba_get_poses_bulk(self._h, self.n_cams, t.reshape(-1), R.reshape(-1)))
        return t, R

    def GetPointsBulk(self):
        p = np.zeros((self.n_pts, 3), np.float32)
        check(self._L.mage_ba_get_points_bulk(self._h, self.n_pts, p.reshape(-1)))
        return p

    # --- diagnostics
    def poses_f64(self):
        out = np.zeros((self.n_cams, 7)); pts = np.zeros((self.n_pts, 3))
        check(self._L.mage_ba_get_state_f64(self._h, out.reshape(-1), pts.reshape(-1)))
        return out

    def points_f64(self):
        out = np.zeros((self.n_cams, 7)); pts = np.zeros((self.n_pts, 3))
        check(self._L.mage_ba_get_state_f64(self._h, out.reshape(-1), pts.reshape(-1)))
        return pts

    def trace(self):
        arr = getattr(self, "_stats_buf", None)
        if arr is None:
            arr = self._stats_buf = (IterStats * 64)()
        n = C.c_size_t(0)
        check(self._L.mage_ba_get_iter_stats(self._h, arr, 64, C.byref(n)))
        return [dict(code=a.code, trials=a.trials, chi_before=a.chi2_before, chi_after=a.chi2_after, lam=a.lambda_)
                for a in arr[: n.value]]

    def use_skyline(self, on=True):
        """The dense solve skips the tiles left of the reduced system's skyline (same numbers, less work); from the next structure build on."""
        check(self._L.mage_ba_use_skyline(self._h, int(bool(on))))

    def schur_per_block(self, on=True):
        """A/B: the Schur build as one wavefront per block (rounds 2-5) instead of the resident stream kernel; the same bits."""
        check(self._L.mage_ba_debug_schur_per_block(self._h, int(bool(on))))

    def enable_profiling(self, on=True):
        """True / 1: every stage of an LM iteration bracketed by HIP events; 2: only the dense factorisation + solves; False: off."""
        check(self._L.mage_ba_enable_profiling(self._h, int(on)))

    STRUCTURE_LISTS = ("cam2hc", "hc2cam", "L_edge", "L_uv", "L_info", "L_cam", "L_pt", "L_slot", "lm_ptr", "lm_pt", "lm_wptr", "w_hc",
                       "w_lm", "camE_ptr", "camE", "camS_ptr", "camS", "blk_ptr", "blk_ij", "con", "blk_order", "stream_ptr", "stream_blks")

    def structure(self, name: str) -> np.ndarray:
        """mage_ba_debug_structure: one list of the graph structure as it sits in HBM (int32 / uint32 views; "L_uv", "L_info" float32;
        "sizes": n_L, n_lm, n_fc, n_w, n_blk, n_blk_slots, n_con, dup_slots, n_pad, built_on_device)."""
        n = C.c_size_t(0)
        check(self._L.mage_ba_debug_structure(self._h, name.encode(), None, 0, C.byref(n)))
        buf = np.zeros(max(n.value, 1), np.uint8)
        check(self._L.mage_ba_debug_structure(self._h, name.encode(), buf.ctypes.data, buf.size, C.byref(n)))
        raw = buf[: n.value]
        return raw.view(np.float32) if name in ("L_uv", "L_info") else raw.view(np.int32)

    def profile(self) -> Profile:
        p = Profile()
        check(self._L.mage_ba_get_profile(self._h, C.byref(p)))
        return p


def load_scene(bundler, scene, bulk: bool = False) -> None:
    """Feed a scene through the BundlerLib call protocol (order of BundleAdjust.cpp:25-193)."""
    bundler.AllocateCameras(scene.n_cams)
    Rcm = scene.cam_R_colmajor()
    if bulk:
        bundler.SetCamerasBulk(scene.cam_t, Rcm, scene.cam_K, scene.cam_fixed.astype(np.uint8))
        bundler.AllocateMapPoints(scene.n_pts)
        bundler.SetMapPointsBulk(scene.points)
        bundler.AllocateObservations(scene.n_obs)
        bundler.SetObservationsBulk(scene.obs_uv, scene.obs_cam, scene.obs_pt, scene.obs_info)
        _feed_tethers(bundler, scene)
        return
    for i in range(scene.n_cams):
        bundler.SetCameraPose(i, scene.cam_t[i], Rcm[i], scene.cam_K[i], bool(scene.cam_fixed[i]))
    bundler.AllocateMapPoints(scene.n_pts)
    for i in range(scene.n_pts):
        bundler.SetMapPoint(i, scene.points[i])
    bundler.AllocateObservations(scene.n_obs)
    for i in range(scene.n_obs):
        bundler.SetObservation(i, scene.obs_uv[i], scene.obs_cam[i], scene.obs_pt[i], scene.obs_info[i])
    _feed_tethers(bundler, scene)


def _feed_tethers(bundler, scene) -> None:
    if getattr(scene, "tethers", None) is not None:
        from .scene import feed_tethers
        feed_tethers(bundler, scene.tethers)
