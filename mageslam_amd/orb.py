"""ctypes bindings of the ORB front-end and the brute-force matcher (include/mage_orb.h, include/mage_match.h).

``OrbDetector`` mirrors the reference's class of the same name (Image/OpenCVModified.h:64-173): the constructor takes
the FeatureExtractorSettings scalars, ``DetectAndCompute`` returns keypoints (cv::KeyPoint records) and 32-byte
descriptors.  ``Match`` mirrors Tracking/FeatureMatcher.h:100-109.  These are bindings only; all work is in the shared library.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4")])      # cv::KeyPoint, 28 bytes
DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])   # cv::DMatch

_u8 = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
_i32 = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class BowTree(C.Structure):      # include/mage_match.h mage_bow_tree
    _fields_ = [("node_descriptors", C.c_void_p), ("child_offsets", C.c_void_p), ("children", C.c_void_p), ("n_nodes", C.c_int32)]


class OrbParams(C.Structure):
    _fields_ = [("gaussian_kernel_size", C.c_uint), ("nfeatures", C.c_uint), ("scale_factor", C.c_float), ("nlevels", C.c_uint),
                ("patch_size", C.c_uint), ("fast_threshold", C.c_uint), ("use_orientation", C.c_int), ("feature_factor_anms", C.c_float),
                ("feature_strength_anms", C.c_float), ("strong_response_anms", C.c_int), ("min_robust_factor", C.c_float),
                ("max_robust_factor", C.c_float), ("num_cells_x", C.c_int), ("num_cells_y", C.c_int), ("device", C.c_int)]


class OrbProfile(C.Structure):
    _fields_ = [("fast_ms", C.c_double), ("select_ms", C.c_double), ("blur_ms", C.c_double), ("brief_ms", C.c_double),
                ("total_ms", C.c_double), ("n_frames", C.c_int)]


_declared = False


def _declare():
    global _declared
    if _declared:
        return
    L = lib()
    vp = C.c_void_p
    L.mage_orb_default_params.argtypes = [C.POINTER(OrbParams)]
    L.mage_orb_create.argtypes = [C.POINTER(OrbParams), C.POINTER(vp)]
    L.mage_orb_destroy.argtypes = [vp]; L.mage_orb_destroy.restype = None
    L.mage_orb_detect.argtypes = [vp, _u8, C.c_int, C.c_int, C.c_int, vp, _u8, C.c_int, C.POINTER(C.c_int)]
    L.mage_orb_detect_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, vp, _u8, C.c_int, _i32]
    L.mage_orb_detect_batch_device.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.mage_orb_debug_read.argtypes = [vp, vp, vp]
    L.mage_orb_get_profile.argtypes = [vp, C.POINTER(OrbProfile)]
    L.mage_orb_enable_profile.argtypes = [vp, C.c_int]
    L.mage_orb_undistort_keypoints.argtypes = [vp, vp, C.c_int, C.POINTER(UndistortParams)]
    L.mage_orb_undistort_keypoints_device.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.POINTER(UndistortParams)]
    L.mage_hamming256.argtypes = [_u8, _u8]; L.mage_hamming256.restype = C.c_int
    L.mage_matcher_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.mage_matcher_destroy.argtypes = [vp]; L.mage_matcher_destroy.restype = None
    L.mage_match_bf.argtypes = [vp, _u8, C.c_int, _u8, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.POINTER(C.c_int)]
    L.mage_match_masked.argtypes = [vp, _u8, C.c_int, vp, _u8, C.c_int, vp, C.c_int, C.c_int, vp, C.c_int, C.POINTER(C.c_int)]
    L.mage_match_bf_batch.argtypes = [vp, C.c_int, _u8, _i32, C.c_int, _u8, _i32, C.c_int, C.c_int, C.c_int, vp, C.c_int, _i32]
    L.mage_match_bf_batch_device.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(vp)]
    L.mage_match_radius.argtypes = [vp, vp, C.c_int, vp, vp, _u8, vp, C.c_int, vp, _u8, C.c_float, C.c_int, C.c_int, vp, C.c_int, C.POINTER(C.c_int)]
    L.mage_bow_find_leaf_batch.argtypes = [vp, vp, vp, C.c_int, vp]
    L.mage_bow_set_tree.argtypes = [vp, vp]
    L.mage_match_indexed_bow.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.POINTER(C.c_int)]
    L.mage_match_indexed.argtypes = [vp, _u8, C.c_int, vp, vp, vp, _u8, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.POINTER(C.c_int)]
    L.mage_matcher_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_double)]
    _declared = True


def default_params(**kw) -> OrbParams:
    _declare()
    p = OrbParams()
    check(lib().mage_orb_default_params(C.byref(p)))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class UndistortParams(C.Structure):
    """mage_undistort_params: camera matrix (row-major 3x3), k1 k2 p1 p2 k3 [k4 k5 k6], count, new camera matrix."""
    _fields_ = [("camera_matrix", C.c_float * 9), ("dist_coeffs", C.c_float * 8), ("n_dist", C.c_int), ("new_camera_matrix", C.c_float * 9)]

    @classmethod
    def make(cls, K, dist, P):
        u = cls()
        u.camera_matrix[:] = [float(v) for v in np.asarray(K, np.float32).reshape(9)]
        d = [float(v) for v in np.asarray(dist, np.float32).reshape(-1)]
        u.dist_coeffs[:] = (d + [0.0] * 8)[:8]
        u.n_dist = len(d)
        u.new_camera_matrix[:] = [float(v) for v in np.asarray(P, np.float32).reshape(9)]
        return u


class OrbDetector:
    def __init__(self, params: OrbParams = None, **kw):
        _declare()
        self._L = lib()
        self.params = params or default_params(**kw)
        self._h = C.c_void_p()
        check(self._L.mage_orb_create(C.byref(self.params), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self._L.mage_orb_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def DetectAndCompute(self, image: np.ndarray, capacity: int = None):
        """One CV_8UC1 image -> (keypoints[KEYPOINT_DTYPE], descriptors[n, 32])."""
        img = np.ascontiguousarray(image, np.uint8)
        h, w = img.shape
        cap = int(self.params.nfeatures) if capacity is None else capacity
        kps = np.zeros(max(cap, 1), KEYPOINT_DTYPE); desc = np.zeros((max(cap, 1), 32), np.uint8)
        n = C.c_int(0)
        check(self._L.mage_orb_detect(self._h, img.reshape(-1), w, h, w, kps.ctypes.data_as(C.c_void_p), desc.reshape(-1), cap, C.byref(n)))
        return kps[: n.value].copy(), desc[: n.value].copy()

    def DetectAndComputeBatch(self, images: np.ndarray, capacity: int = None):
        """images: (n, h, w) uint8 -> (keypoints[n, cap], descriptors[n, cap, 32], counts[n])."""
        imgs = np.ascontiguousarray(images, np.uint8)
        n, h, w = imgs.shape
        cap = int(self.params.nfeatures) if capacity is None else capacity
        kps = np.zeros((n, max(cap, 1)), KEYPOINT_DTYPE); desc = np.zeros((n, max(cap, 1), 32), np.uint8); cnt = np.zeros(max(n, 1), np.int32)
        check(self._L.mage_orb_detect_batch(self._h, imgs.ctypes.data_as(C.c_void_p), 0, n, w, h, w, w * h, kps.ctypes.data_as(C.c_void_p),
                                            desc.reshape(-1), cap, cnt))
        return kps, desc, cnt[:n]

    def detect_batch_device(self, images_ptr: int, n: int, w: int, h: int, capacity: int = None):
        """Device-resident in and out (raw device pointers as ints): returns (kp_ptr, desc_ptr, counts_ptr)."""
        cap = int(self.params.nfeatures) if capacity is None else capacity
        kp, de, cn = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(self._L.mage_orb_detect_batch_device(self._h, C.c_void_p(images_ptr), n, w, h, w, w * h, cap, C.byref(kp), C.byref(de), C.byref(cn)))
        return kp.value, de.value, cn.value

    def UndistortKeypoints(self, keypoints: np.ndarray, params: "UndistortParams") -> np.ndarray:
        """OrbFeatureDetector::UndistortKeypoints (OrbFeatureDetector.cpp:30-62) on a copy of a KEYPOINT_DTYPE array."""
        out = np.ascontiguousarray(keypoints, KEYPOINT_DTYPE).copy()
        check(self._L.mage_orb_undistort_keypoints(self._h, out.ctypes.data_as(C.c_void_p) if len(out) else None, len(out), C.byref(params)))
        return out

    def undistort_device(self, kp_ptr: int, counts_ptr: int, n_frames: int, capacity: int, params: "UndistortParams") -> None:
        """In place on the device buffers detect_batch_device returned."""
        check(self._L.mage_orb_undistort_keypoints_device(self._h, C.c_void_p(kp_ptr), C.c_void_p(counts_ptr), n_frames, capacity, C.byref(params)))

    def debug_read(self, w: int, h: int):
        s = np.zeros((h, w), np.uint8); b = np.zeros((h, w), np.uint8)
        check(self._L.mage_orb_debug_read(self._h, s.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)))
        return s, b

    def enable_profile(self, on: bool = True) -> None:
        """Stage timings (HIP events between the kernels) are recorded only after this; off by default."""
        check(self._L.mage_orb_enable_profile(self._h, int(on)))

    def profile(self) -> OrbProfile:
        p = OrbProfile()
        check(self._L.mage_orb_get_profile(self._h, C.byref(p)))
        return p


def GetDescriptorDistance(d0: np.ndarray, d1: np.ndarray) -> int:
    _declare()
    return int(lib().mage_hamming256(np.ascontiguousarray(d0, np.uint8).reshape(32), np.ascontiguousarray(d1, np.uint8).reshape(32)))


class Matcher:
    def __init__(self, device: int = -1):
        _declare()
        self._L = lib()
        self._h = C.c_void_p()
        check(self._L.mage_matcher_create(device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            self._L.mage_matcher_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def Match(self, descA, descB, maskA=None, maskB=None, max_hamming_dist=30, min_hamming_difference=1) -> np.ndarray:
        """FeatureMatcher::Match: returns DMATCH_DTYPE records with ORIGINAL indices, ascending in A."""
        A = np.ascontiguousarray(descA, np.uint8).reshape(-1, 32); B = np.ascontiguousarray(descB, np.uint8).reshape(-1, 32)
        out = np.zeros(max(len(A), 1), DMATCH_DTYPE)
        n = C.c_int(0)
        ma = None if maskA is None else np.ascontiguousarray(maskA, np.uint8)
        mb = None if maskB is None else np.ascontiguousarray(maskB, np.uint8)
        a_arr = A.reshape(-1) if len(A) else np.zeros(32, np.uint8)
        b_arr = B.reshape(-1) if len(B) else np.zeros(32, np.uint8)
        check(self._L.mage_match_masked(self._h, a_arr, len(A), None if ma is None else ma.ctypes.data_as(C.c_void_p), b_arr, len(B),
                                        None if mb is None else mb.ctypes.data_as(C.c_void_p), int(max_hamming_dist),
                                        int(min_hamming_difference), out.ctypes.data_as(C.c_void_p), len(out), C.byref(n)))
        return out[: min(n.value, len(out))].copy()

    def MatchBatch(self, descA, countsA, descB, countsB, max_hamming_dist=30, min_hamming_difference=1):
        """descA: (n_pairs, capA, 32), descB: (n_pairs, capB, 32) -> (matches[n_pairs, capA], counts[n_pairs])."""
        A = np.ascontiguousarray(descA, np.uint8); B = np.ascontiguousarray(descB, np.uint8)
        npairs, capA, _ = A.shape; capB = B.shape[1]
        out = np.zeros((npairs, max(capA, 1)), DMATCH_DTYPE); cnt = np.zeros(max(npairs, 1), np.int32)
        check(self._L.mage_match_bf_batch(self._h, npairs, A.reshape(-1), np.ascontiguousarray(countsA, np.int32), capA, B.reshape(-1),
                                          np.ascontiguousarray(countsB, np.int32), capB, int(max_hamming_dist), int(min_hamming_difference),
                                          out.ctypes.data_as(C.c_void_p), capA, cnt))
        return out, cnt[:npairs]

    def match_batch_device(self, n_pairs, dA, cA, capA, dB, cB, capB, max_hamming_dist=30, min_hamming_difference=1, cap_out=None):
        o, c = C.c_void_p(), C.c_void_p()
        check(self._L.mage_match_bf_batch_device(self._h, n_pairs, C.c_void_p(dA), C.c_void_p(cA), capA, C.c_void_p(dB), C.c_void_p(cB), capB,
                                                 int(max_hamming_dist), int(min_hamming_difference), capA if cap_out is None else cap_out,
                                                 C.byref(o), C.byref(c)))
        return o.value, c.value

    def RadiusMatch(self, query_keypoints, query_descriptors, target_keypoints, target_descriptors, radius, max_hamming_dist=30,
                    min_hamming_difference=1, query_position_overrides=None, query_mask=None, target_mask=None) -> np.ndarray:
        """FeatureMatcher::RadiusMatch (multi-query form, FeatureMatcher.cpp:294-378); keypoints are KEYPOINT_DTYPE arrays."""
        qk = np.ascontiguousarray(query_keypoints, KEYPOINT_DTYPE); tk = np.ascontiguousarray(target_keypoints, KEYPOINT_DTYPE)
        qd = np.ascontiguousarray(query_descriptors, np.uint8).reshape(-1); td = np.ascontiguousarray(target_descriptors, np.uint8).reshape(-1)
        if qd.size == 0: qd = np.zeros(32, np.uint8)
        if td.size == 0: td = np.zeros(32, np.uint8)
        ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        qp = None if query_position_overrides is None else np.ascontiguousarray(query_position_overrides, np.float32)
        qm = None if query_mask is None else np.ascontiguousarray(query_mask, np.uint8)
        tm = None if target_mask is None else np.ascontiguousarray(target_mask, np.uint8)
        out = np.zeros(max(len(qk), 1), DMATCH_DTYPE)
        n = C.c_int(0)
        check(self._L.mage_match_radius(self._h, ptr(qk), len(qk), ptr(qp), ptr(qm), qd, ptr(tk), len(tk), ptr(tm), td, float(radius),
                                        int(max_hamming_dist), int(min_hamming_difference), ptr(out), len(out), C.byref(n)))
        return out[: min(n.value, len(out))].copy()

    def IndexedMatch(self, descriptors_a, cand_b_offsets, cand_b, descriptors_b, cand_a_offsets, cand_a, max_hamming_dist=30,
                     min_hamming_difference=1, mask_a=None, mask_b=None) -> np.ndarray:
        """FeatureMatcher::IndexedMatch (FeatureMatcher.cpp:192-292) with the vocabulary index's candidate lists in CSR form
        (cand_b: indices into B per A descriptor; cand_a: indices into A per B descriptor, for the reverse check)."""
        A = np.ascontiguousarray(descriptors_a, np.uint8).reshape(-1); B = np.ascontiguousarray(descriptors_b, np.uint8).reshape(-1)
        nA, nB = A.size // 32, B.size // 32
        if A.size == 0: A = np.zeros(32, np.uint8)
        if B.size == 0: B = np.zeros(32, np.uint8)
        ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        bo = np.ascontiguousarray(cand_b_offsets, np.int32); bc = np.ascontiguousarray(cand_b, np.int32)
        ao = np.ascontiguousarray(cand_a_offsets, np.int32); ac = np.ascontiguousarray(cand_a, np.int32)
        ma = None if mask_a is None else np.ascontiguousarray(mask_a, np.uint8)
        mb = None if mask_b is None else np.ascontiguousarray(mask_b, np.uint8)
        out = np.zeros(max(nA, 1), DMATCH_DTYPE)
        n = C.c_int(0)
        check(self._L.mage_match_indexed(self._h, A, nA, ptr(ma), ptr(bo), ptr(bc) if bc.size else None, B, nB, ptr(mb), ptr(ao), ptr(ac) if ac.size else None,
                                         int(max_hamming_dist), int(min_hamming_difference), ptr(out), len(out), C.byref(n)))
        return out[: min(n.value, len(out))].copy()

    @staticmethod
    def _bow_tree(node_descriptors, child_offsets, children):
        nd = np.ascontiguousarray(node_descriptors, np.uint8).reshape(-1, 32)
        co = np.ascontiguousarray(child_offsets, np.int32); ch = np.ascontiguousarray(children, np.int32)
        if ch.size == 0: ch = np.zeros(1, np.int32)
        t = BowTree(nd.ctypes.data_as(C.c_void_p), co.ctypes.data_as(C.c_void_p), ch.ctypes.data_as(C.c_void_p), len(nd))
        return t, (nd, co, ch)          # (the arrays must outlive the call)

    def BowSetTree(self, node_descriptors=None, child_offsets=None, children=None) -> None:
        """mage_bow_set_tree: keeps a validated copy of the tree on the device; BowFindLeaf / IndexedMatchBow called with node_descriptors=None
        use it.  Without arguments: forgets it."""
        if node_descriptors is None:
            check(self._L.mage_bow_set_tree(self._h, None))
            return
        t, keep = self._bow_tree(node_descriptors, child_offsets, children)
        check(self._L.mage_bow_set_tree(self._h, C.byref(t)))

    def BowFindLeaf(self, node_descriptors, child_offsets, children, descriptors) -> np.ndarray:
        """OnlineBow::FindLeafNode (BoW/OnlineBow.cpp:289-311) for a batch of descriptors; the tree as flat arrays (include/mage_match.h),
        or node_descriptors=None for the tree kept by BowSetTree."""
        t, keep = (None, None) if node_descriptors is None else self._bow_tree(node_descriptors, child_offsets, children)
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        leaf = np.zeros(max(len(d), 1), np.int32)
        check(self._L.mage_bow_find_leaf_batch(self._h, None if t is None else C.byref(t), d.ctypes.data_as(C.c_void_p) if len(d) else None, len(d), leaf.ctypes.data_as(C.c_void_p)))
        return leaf[:len(d)]

    def IndexedMatchBow(self, node_descriptors, child_offsets, children, descriptors_a, leaf_features_a_offsets, leaf_features_a, descriptors_b,
                        leaf_features_b_offsets, leaf_features_b, max_hamming_dist=30, min_hamming_difference=1, mask_a=None, mask_b=None) -> np.ndarray:
        """IndexedMatch (FeatureMatcher.cpp:192-292) with the candidate lists looked up in the vocabulary tree on the device
        (OnlineBow::QueryFeatures, BoW/OnlineBow.cpp:115-132): leaf_features_x = per node the features of image x filed under it (CSR)."""
        t, keep = (None, None) if node_descriptors is None else self._bow_tree(node_descriptors, child_offsets, children)
        A = np.ascontiguousarray(descriptors_a, np.uint8).reshape(-1); B = np.ascontiguousarray(descriptors_b, np.uint8).reshape(-1)
        nA, nB = A.size // 32, B.size // 32
        if A.size == 0: A = np.zeros(32, np.uint8)
        if B.size == 0: B = np.zeros(32, np.uint8)
        ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        fao = np.ascontiguousarray(leaf_features_a_offsets, np.int32); fa = np.ascontiguousarray(leaf_features_a, np.int32)
        fbo = np.ascontiguousarray(leaf_features_b_offsets, np.int32); fb = np.ascontiguousarray(leaf_features_b, np.int32)
        ma = None if mask_a is None else np.ascontiguousarray(mask_a, np.uint8)
        mb = None if mask_b is None else np.ascontiguousarray(mask_b, np.uint8)
        out = np.zeros(max(nA, 1), DMATCH_DTYPE)
        n = C.c_int(0)
        check(self._L.mage_match_indexed_bow(self._h, None if t is None else C.byref(t), ptr(A), nA, ptr(ma), ptr(fao), ptr(fa) if fa.size else None, ptr(B), nB, ptr(mb), ptr(fbo),
                                             ptr(fb) if fb.size else None, int(max_hamming_dist), int(min_hamming_difference), ptr(out), len(out), C.byref(n)))
        return out[: min(n.value, len(out))].copy()

    def last_kernel_ms(self) -> float:
        v = C.c_double(0)
        check(self._L.mage_matcher_last_kernel_ms(self._h, C.byref(v)))
        return float(v.value)
