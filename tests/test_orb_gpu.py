"""GPU parity tests (-m gpu) of the ORB front-end and the brute-force matcher, through the C ABI.
Every output is integer / byte data: the comparison is bit-exact against the golden fixtures and the CPU oracle."""
import hashlib
import os

import numpy as np
import pytest

from mageslam_amd import frames
from mageslam_amd.orb import GetDescriptorDistance, Matcher, OrbDetector, default_params
from oracle import oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "orb_frames.npz")
CASES = ["orb_64x48", "orb_160x120", "orb_640x480_a", "orb_640x480_b"]


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLDEN)


def kp_xyr(k):
    return np.stack([k["x"], k["y"], k["response"]], axis=1).astype(np.int64)


@pytest.mark.parametrize("name", CASES)
def test_detect_matches_golden(gold, name):
    img = gold[name + "_img"]
    det = OrbDetector()
    k, d = det.DetectAndCompute(img)
    assert np.array_equal(kp_xyr(k), gold[name + "_kp"])
    assert np.array_equal(d, gold[name + "_desc"])
    score, blur = det.debug_read(img.shape[1], img.shape[0])
    assert hashlib.sha256(blur.tobytes()).hexdigest() == str(gold[name + "_blursha"])
    ref_score = np.zeros_like(img)
    O.lib().orbo_fast_score_map(img, img.shape[1], img.shape[0], img.shape[1], 4, ref_score)
    assert np.array_equal(score, ref_score)
    assert np.all(k["angle"] == 0) and np.all(k["size"] == 15) and np.all(k["octave"] == 0) and np.all(k["class_id"] == -1)


def test_batch_of_frames_equals_oracle_frame_by_frame():
    imgs = np.stack([frames.make_frame(200 + i, 320, 180) for i in range(6)])       # the Console app's tracking resolution
    det = OrbDetector()
    K, D, C = det.DetectAndComputeBatch(imgs)
    for i in range(len(imgs)):
        k, d = O.orb_detect(imgs[i])
        assert C[i] == len(k)
        assert np.array_equal(kp_xyr(K[i, : C[i]]), kp_xyr(k)) and np.array_equal(D[i, : C[i]], d)


def test_detectors_and_matchers_on_concurrent_threads(gold):
    """One detector per camera, one frame in flight per detector, several cameras at once: four detector +
    matcher handles on their own threads and streams give the fixture results."""
    import threading
    errors = []

    def work(name):
        try:
            det, m = OrbDetector(), Matcher()
            for _ in range(4):
                k, d = det.DetectAndCompute(gold[name + "_img"])
                assert np.array_equal(kp_xyr(k), gold[name + "_kp"]) and np.array_equal(d, gold[name + "_desc"])
                got = m.Match(gold["orb_640x480_a_desc"], gold["orb_640x480_b_desc"], None, None, 30, 1)
                got = np.stack([got["queryIdx"], got["trainIdx"], got["distance"].astype(np.int64)], axis=1)
                assert np.array_equal(got, gold["matches_ab"])
        except BaseException as e:        # noqa: BLE001 - reported to the main thread
            errors.append((name, repr(e)))

    threads = [threading.Thread(target=work, args=(n,)) for n in CASES]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors


@pytest.mark.parametrize("kw", [dict(patch_size=31), dict(gaussian_kernel_size=1), dict(gaussian_kernel_size=5, nfeatures=100),
                                dict(fast_threshold=20, num_cells_x=8, num_cells_y=5), dict(feature_factor_anms=1.0, max_robust_factor=2.2)])
def test_non_default_settings_match_oracle(kw):
    img = frames.make_frame(31, 200, 150, n_rect=40, n_disc=60)
    okw = {"feature_factor_anms": "feature_factor", "feature_strength_anms": "feature_strength", "strong_response_anms": "strong_response",
           "min_robust_factor": "min_robust", "max_robust_factor": "max_robust", "num_cells_x": "cells_x", "num_cells_y": "cells_y"}
    k, d = OrbDetector(**kw).DetectAndCompute(img)
    ko, do = O.orb_detect(img, O.OrbParams.defaults(**{okw.get(a, a): b for a, b in kw.items()}))
    assert np.array_equal(kp_xyr(k), kp_xyr(ko)) and np.array_equal(d, do)


@pytest.mark.parametrize("size", [(640, 480), (333, 241), (70, 50), (64, 24), (25, 31)])
def test_matrix_core_blur_equals_vector_blur_and_oracle(size):
    """The fused 7-tap Gaussian has two forms -- two banded i8 products on the matrix cores (default) and the two-pass v_dot4 form
    (MAGE_ORB_BLUR=valu, read when the detector is created; also the fall-back for taps that do not fit i8): the blurred image
    (mage_orb_debug_read), the FAST score map, the keypoints and the descriptors must be the same bytes, and equal to the oracle's.
    Sizes: full frame, widths / heights that are no multiple of the 64 x 24 tile (partial dwords at the right edge), one tile, and a
    frame smaller than a tile (every row and column reflected)."""
    w, h = size
    img = frames.make_frame(4242 + w, w, h, n_rect=30, n_disc=40)
    kw = dict(nfeatures=200, num_cells_x=8, num_cells_y=6) if w < 640 else {}
    old = os.environ.pop("MAGE_ORB_BLUR", None)
    try:
        dm = OrbDetector(**kw)
        os.environ["MAGE_ORB_BLUR"] = "valu"
        dv = OrbDetector(**kw)
    finally:
        os.environ.pop("MAGE_ORB_BLUR", None)
        if old is not None:
            os.environ["MAGE_ORB_BLUR"] = old
    km, desc_m = dm.DetectAndCompute(img)
    kv, desc_v = dv.DetectAndCompute(img)
    sm, bm = dm.debug_read(w, h)
    sv, bv = dv.debug_read(w, h)
    assert np.array_equal(bm, bv), np.argwhere(bm != bv)[:5]
    assert np.array_equal(sm, sv)
    assert np.array_equal(kp_xyr(km), kp_xyr(kv)) and np.array_equal(desc_m, desc_v)
    okw = {"num_cells_x": "cells_x", "num_cells_y": "cells_y"}
    ko, do, bo = O.orb_detect(img, O.OrbParams.defaults(**{okw.get(a, a): b for a, b in kw.items()}), want_blur=True)
    assert np.array_equal(bm, bo), np.argwhere(bm != bo)[:5]
    assert np.array_equal(kp_xyr(km), kp_xyr(ko)) and np.array_equal(desc_m, do)


@pytest.mark.parametrize("threshold", [1, 2, 127, 128, 200, 254, 255])
def test_extreme_thresholds_and_saturated_pixels_match_oracle(threshold):
    """The two FAST screens compare through guard-bit subtractions on 16-bit fields ((V + 0x8000 - t - 1) - R and (V + 0x8000 + t) - R):
    the fields must neither borrow nor carry at the ends of the ranges -- thresholds up to 255, ring / centre pixels of 0 and 255.
    Frames: black / white blocks and isolated extreme pixels on mid grey, and uniform noise over the full range."""
    rng = np.random.default_rng(77 + threshold)
    w, h = 200, 120
    a = np.full((h, w), 128, np.uint8)
    a[10:40, 10:60] = 0; a[50:90, 30:90] = 255; a[20:30, 100:180] = 255; a[25:27, 120:160] = 0
    for _ in range(150):
        a[int(rng.integers(4, h - 4)), int(rng.integers(4, w - 4))] = int(rng.choice([0, 1, 254, 255]))
    b = rng.integers(0, 256, (h, w)).astype(np.uint8)
    c = (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8)
    for img in (a, b, c):
        kw = dict(fast_threshold=threshold, nfeatures=300, num_cells_x=8, num_cells_y=6)
        det = OrbDetector(**kw)
        k, d = det.DetectAndCompute(img)
        score, _ = det.debug_read(w, h)
        ko, do = O.orb_detect(img, O.OrbParams.defaults(fast_threshold=threshold, nfeatures=300, cells_x=8, cells_y=6))
        assert np.array_equal(kp_xyr(k), kp_xyr(ko)), (threshold, len(k), len(ko))
        assert np.array_equal(d, do)
        if threshold >= 128:
            # the literal definition on the centre pixels the detector scores (3 pixels from the border): a corner needs 9 contiguous
            # ring pixels all > v + t or all < v - t, impossible for t = 255
            assert threshold < 255 or not score.any()


@pytest.mark.parametrize("case", range(24))
def test_randomised_settings_and_frames_match_oracle(case):
    """Differential test over the detector's whole parameter surface: random frame size and content, feature budget, FAST
    threshold, cell grid, ANMS factors, blur kernel, patch 15 / 31, pyramid depth and scale, orientation.  Keypoints (x, y,
    response, octave, size, angle) and descriptors must equal the oracle's bit for bit."""
    rng = np.random.default_rng(0x0B5E + case)
    w, h = int(rng.integers(40, 360)), int(rng.integers(40, 280))
    img = frames.make_frame(900 + case, w, h, n_rect=int(rng.integers(5, 80)), n_disc=int(rng.integers(5, 120)))
    if case % 5 == 4:
        img = (rng.integers(0, 256, (h, w))).astype(np.uint8)                      # pure noise: thousands of raw corners
    nlevels = int(rng.integers(1, 5)) if case % 2 else 1
    kw = dict(nfeatures=int(rng.integers(8, 700)), fast_threshold=int(rng.integers(1, 40)), num_cells_x=int(rng.integers(1, 40)),
              num_cells_y=int(rng.integers(1, 40)), gaussian_kernel_size=int(rng.choice([1, 3, 5, 7, 9])),
              patch_size=int(rng.choice([15, 31])), nlevels=nlevels, scale_factor=float(rng.choice([1.2, 1.5, 2.0])),
              use_orientation=int(case % 3 == 0), feature_factor_anms=float(rng.choice([1.0, 1.5, 2.5])),
              feature_strength_anms=float(rng.choice([0.5, 0.9, 1.0])), strong_response_anms=int(rng.integers(5, 60)),
              min_robust_factor=float(rng.choice([1.0, 1.1])), max_robust_factor=float(rng.choice([2.0, 2.2, 3.0])))
    okw = {"feature_factor_anms": "feature_factor", "feature_strength_anms": "feature_strength", "strong_response_anms": "strong_response",
           "min_robust_factor": "min_robust", "max_robust_factor": "max_robust", "num_cells_x": "cells_x", "num_cells_y": "cells_y"}
    k, d = OrbDetector(**kw).DetectAndCompute(img)
    ko, do = O.orb_detect(img, O.OrbParams.defaults(**{okw.get(a, a): b for a, b in kw.items()}))
    assert len(k) == len(ko), (kw, w, h)
    for f in ("x", "y", "response", "octave", "size", "angle", "class_id"):
        assert np.array_equal(k[f], ko[f]), (f, kw, w, h)
    assert np.array_equal(d, do), (kw, w, h)


def test_edge_cases_match_oracle():
    det = OrbDetector()
    for shape in ((5, 5), (14, 40), (40, 14), (1, 1), (15, 15), (16, 33)):
        k, d = det.DetectAndCompute(np.full(shape, 77, np.uint8))
        assert len(k) == 0
    # many equal responses (checkerboard): ties everywhere in the selection -> the canonical order must still agree
    yy, xx = np.mgrid[0:120, 0:160]
    chk = (((xx // 8) + (yy // 8)) % 2 * 200 + 20).astype(np.uint8)
    k, d = det.DetectAndCompute(chk)
    ko, do = O.orb_detect(chk)
    assert np.array_equal(kp_xyr(k), kp_xyr(ko)) and np.array_equal(d, do)
    # saturated noise: thousands of raw corners
    rng = np.random.default_rng(0)
    noise = rng.integers(0, 2, (240, 320)).astype(np.uint8) * 255
    k, d = det.DetectAndCompute(noise)
    ko, do = O.orb_detect(noise)
    assert np.array_equal(kp_xyr(k), kp_xyr(ko)) and np.array_equal(d, do)
    # capacity truncation
    a = frames.make_frame(7)
    k, d = det.DetectAndCompute(a, capacity=100)
    ko, do = O.orb_detect(a, cap=100)
    assert len(k) == 100 and np.array_equal(kp_xyr(k), kp_xyr(ko)) and np.array_equal(d, do)


@pytest.mark.parametrize("shape, kw", [
    ((1200, 1600), dict(nfeatures=3000)),                                   # 25 x 50 FAST tiles (> 1024: several per thread of the selection), working set beyond LDS
    ((1200, 1600), dict(nfeatures=3000, feature_strength_anms=1.0, feature_factor_anms=1.0)),
    ((700, 1100), dict(nfeatures=2500, num_cells_x=40, num_cells_y=40)),    # more grid cells than the LDS index holds
    ((480, 640), dict(nfeatures=5000)),                                     # fewer detections than the quota: everything is kept, in raster order
    ((60, 400), dict(nfeatures=30, num_cells_x=64, num_cells_y=64)),        # bounding box narrower than the grid in y: the reference's ring walk, step by step
])
def test_selection_fallback_paths_match_oracle(shape, kw):
    h, w = shape
    img = frames.make_frame(0x51 + h, width=w, height=h)
    okw = {"feature_factor_anms": "feature_factor", "feature_strength_anms": "feature_strength", "num_cells_x": "cells_x", "num_cells_y": "cells_y"}
    k, d = OrbDetector(**kw).DetectAndCompute(img)
    ko, do = O.orb_detect(img, O.OrbParams.defaults(**{okw.get(a, a): b for a, b in kw.items()}))
    assert len(k) == len(ko) and len(k) > 0
    assert np.array_equal(kp_xyr(k), kp_xyr(ko)) and np.array_equal(d, do)


def test_unsupported_settings_are_refused():
    from mageslam_amd._lib import MageError
    for kw in (dict(nlevels=17), dict(nlevels=2, scale_factor=1.0), dict(patch_size=200)):
        with pytest.raises(MageError):
            OrbDetector(**kw)


def test_match_golden_and_oracle(gold):
    A, B = gold["orb_640x480_a_desc"], gold["orb_640x480_b_desc"]
    mt = Matcher()
    m = mt.Match(A, B, None, None, 30, 1)
    got = np.stack([m["queryIdx"], m["trainIdx"], m["distance"].astype(np.int64)], axis=1)
    assert np.array_equal(got, gold["matches_ab"]) and np.all(m["imgIdx"] == -1)
    for md, mdiff in ((40, 2), (20, 1), (256, 0), (0, 1)):
        m = mt.Match(A, B, None, None, md, mdiff)
        mo = O.match(A, B, md, mdiff)
        assert np.array_equal(m, mo)


def test_match_masks_map_back_to_original_indices(gold):
    A, B = gold["orb_640x480_a_desc"], gold["orb_640x480_b_desc"]
    rng = np.random.default_rng(1)
    ma = rng.random(len(A)) < 0.7; mb = rng.random(len(B)) < 0.6
    m = Matcher().Match(A, B, ma, mb, 30, 1)
    ia, ib = np.nonzero(ma)[0], np.nonzero(mb)[0]
    mo = O.match(A[ia], B[ib], 30, 1)
    assert np.array_equal(m["queryIdx"], ia[mo["queryIdx"]]) and np.array_equal(m["trainIdx"], ib[mo["trainIdx"]])
    assert np.array_equal(m["distance"], mo["distance"])
    assert len(Matcher().Match(A, B, np.zeros(len(A), bool), None)) == 0      # empty side (FeatureMatcher.cpp:72-77)


def test_match_batch_ragged_pairs():
    rng = np.random.default_rng(2)
    npairs, cap = 5, 64
    A = rng.integers(0, 256, (npairs, cap, 32)).astype(np.uint8)
    B = A.copy()
    flips = rng.integers(0, 256, B.shape).astype(np.uint8) & (rng.random(B.shape) < 0.03).astype(np.uint8) * 255
    B ^= flips
    B = B[:, ::-1].copy()                                   # shuffled order
    cA = np.array([64, 10, 0, 33, 64], np.int32); cB = np.array([64, 64, 20, 0, 7], np.int32)
    out, cnt = Matcher().MatchBatch(A, cA, B, cB, 60, 1)
    for p in range(npairs):
        mo = O.match(A[p, : cA[p]], B[p, : cB[p]], 60, 1)
        assert cnt[p] == len(mo) and np.array_equal(out[p, : cnt[p]], mo)


def test_match_few_pairs_and_many_pairs_take_different_kernels_same_records():
    """Below 256 pairs a pair is spread over several workgroups (k_match_rows: lane per row, last workgroup finishes), from 256
    on one workgroup takes a pair (k_match).  320 ragged pairs through the second, their first 7 through the first, every pair
    against the oracle; sets larger than one row group (64) and near-duplicate descriptors so that ties for the best occur."""
    rng = np.random.default_rng(77)
    npairs, cap = 320, 96
    centres = rng.integers(0, 256, (6, 32)).astype(np.uint8)
    A = centres[rng.integers(0, 6, (npairs, cap))].copy()
    A ^= np.packbits((rng.random((npairs, cap, 32, 8)) < 0.02).astype(np.uint8), axis=3).reshape(npairs, cap, 32)
    B = A[:, ::-1].copy()
    B ^= np.packbits((rng.random((npairs, cap, 32, 8)) < 0.01).astype(np.uint8), axis=3).reshape(npairs, cap, 32)
    cA = rng.integers(0, cap + 1, npairs).astype(np.int32); cB = rng.integers(0, cap + 1, npairs).astype(np.int32)
    cA[:3] = [cap, 0, 65]; cB[:3] = [cap, 40, 1]
    mt = Matcher()
    out_many, cnt_many = mt.MatchBatch(A, cA, B, cB, 40, 1)
    out_few, cnt_few = mt.MatchBatch(A[:7], cA[:7], B[:7], cB[:7], 40, 1)
    for p in range(npairs):
        mo = O.match(A[p, : cA[p]], B[p, : cB[p]], 40, 1)
        assert cnt_many[p] == len(mo) and np.array_equal(out_many[p, : cnt_many[p]], mo), p
        if p < 7:
            assert cnt_few[p] == len(mo) and np.array_equal(out_few[p, : cnt_few[p]], mo), p
    # the arrival counters are back at zero: a second call on the same handle gives the same records
    out2, cnt2 = mt.MatchBatch(A[:7], cA[:7], B[:7], cB[:7], 40, 1)
    assert np.array_equal(cnt2, cnt_few) and all(np.array_equal(out2[p, : cnt2[p]], out_few[p, : cnt_few[p]]) for p in range(7))


@pytest.mark.parametrize("case", range(16))
def test_randomised_descriptor_sets_match_oracle(case):
    """Brute-force matcher on adversarial sets: clusters of near-identical descriptors (ties between best and second best,
    between queries claiming the same train descriptor), exact duplicates, empty and one-element sides, random masks and
    thresholds including 0 and 256.  The match list must equal the oracle's record for record."""
    rng = np.random.default_rng(0x3A7C + case)
    nA, nB = int(rng.integers(0, 300)), int(rng.integers(0, 300))
    if case == 0: nA = 0
    if case == 1: nB = 1
    centres = rng.integers(0, 256, (max(1, int(rng.integers(1, 12))), 32)).astype(np.uint8)

    def draw(n):
        d = centres[rng.integers(0, len(centres), n)].copy()
        noise = (rng.random((n, 32, 8)) < rng.choice([0.0, 0.01, 0.05])).astype(np.uint8)
        d ^= np.packbits(noise, axis=2).reshape(n, 32)
        fresh = rng.random(n) < 0.3
        d[fresh] = rng.integers(0, 256, (int(fresh.sum()), 32)).astype(np.uint8)
        return d

    A, B = draw(nA), draw(nB)
    md, mdiff = int(rng.choice([0, 10, 30, 64, 256])), int(rng.choice([0, 1, 2, 5]))
    ma = rng.random(nA) < 0.8 if case % 2 else None
    mb = rng.random(nB) < 0.8 if case % 3 == 0 else None
    m = Matcher().Match(A, B, ma, mb, md, mdiff)
    ia = np.arange(nA) if ma is None else np.nonzero(ma)[0]
    ib = np.arange(nB) if mb is None else np.nonzero(mb)[0]
    mo = O.match(A[ia], B[ib], md, mdiff)
    assert len(m) == len(mo)
    if len(mo):
        assert np.array_equal(m["queryIdx"], ia[mo["queryIdx"]]) and np.array_equal(m["trainIdx"], ib[mo["trainIdx"]])
        assert np.array_equal(m["distance"], mo["distance"])


def test_hamming_distance_helper():
    rng = np.random.default_rng(4)
    a = rng.integers(0, 256, 32).astype(np.uint8); b = rng.integers(0, 256, 32).astype(np.uint8)
    assert GetDescriptorDistance(a, b) == sum(bin(int(x) ^ int(y)).count("1") for x, y in zip(a, b))


def test_full_size_properties():
    """BASELINE.json configs[1] at 640x480: determinism run to run, descriptors of an image equal descriptors of the same
    image embedded with a different row pitch, match(A, A) with minDiff 0 is the identity wherever descriptors are unique."""
    a, b = frames.frame_pair(11)
    det = OrbDetector()
    k1, d1 = det.DetectAndCompute(a)
    k2, d2 = det.DetectAndCompute(a)
    assert np.array_equal(k1, k2) and np.array_equal(d1, d2)
    assert len(k1) == 440
    m = Matcher().Match(d1, d1, None, None, 0, 1)
    uniq = np.array([np.sum(np.all(d1 == d1[i], axis=1)) == 1 for i in range(len(d1))])
    assert np.array_equal(m["queryIdx"], m["trainIdx"]) and set(m["queryIdx"]) == set(np.nonzero(uniq)[0])


def _kp_array(xyr):
    k = np.zeros(len(xyr), O.KEYPOINT_DTYPE)
    k["x"], k["y"], k["response"], k["size"], k["class_id"] = xyr[:, 0], xyr[:, 1], xyr[:, 2], 15, -1
    return k


def test_radius_match_golden_and_oracle(gold):
    """SURVEY.md 8f rank 2 ("next" row M-5): RadiusMatch, bit-exact against the fixture and the oracle."""
    ka, kb = _kp_array(gold["orb_640x480_a_kp"]), _kp_array(gold["orb_640x480_b_kp"])
    da, db = gold["orb_640x480_a_desc"], gold["orb_640x480_b_desc"]
    as3 = lambda m: np.stack([m["queryIdx"], m["trainIdx"], m["distance"].astype(np.int64)], axis=1)
    mt = Matcher()
    m = mt.RadiusMatch(ka, da, kb, db, 20.0, 30, 1)
    assert np.array_equal(as3(m), gold["radius_plain"]) and np.all(m["imgIdx"] == 0)
    m = mt.RadiusMatch(ka, da, kb, db, 4.0, 40, 2, query_position_overrides=gold["radius_qpos"])
    assert np.array_equal(as3(m), gold["radius_override"])
    rng = np.random.default_rng(8)
    for radius, md, mdiff in ((1.0, 30, 1), (35.5, 60, 0), (640.0, 256, 3), (0.0, 30, 1)):
        qm = rng.random(len(ka)) < 0.8; tm = rng.random(len(kb)) < 0.8
        m = mt.RadiusMatch(ka, da, kb, db, radius, md, mdiff, query_mask=qm, target_mask=tm)
        mo = O.radius_match(ka, da, kb, db, radius, md, mdiff, qmask=qm, tmask=tm)
        assert np.array_equal(m, mo)
    assert len(mt.RadiusMatch(ka[:0], da[:0], kb, db, 5.0)) == 0 and len(mt.RadiusMatch(ka, da, kb[:0], db[:0], 5.0)) == 0
    # a single accepted query skips the uniqueness pass (FeatureMatcher.cpp:374-377)
    m = mt.RadiusMatch(ka[:1], da[:1], kb, db, 30.0, 256, 0)
    assert np.array_equal(m, O.radius_match(ka[:1], da[:1], kb, db, 30.0, 256, 0))


def test_indexed_match_golden_and_oracle(gold):
    """SURVEY.md 8f rank 4 ("next" row M-4): IndexedMatch over caller-supplied candidate lists, bit-exact against the fixture,
    the oracle on random masks / thresholds, and its argument checks."""
    import os
    from mageslam_amd._lib import MageError
    ix = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "orb_indexed.npz"))
    da, db = gold["orb_640x480_a_desc"], gold["orb_640x480_b_desc"]
    as3 = lambda m: np.stack([m["queryIdx"], m["trainIdx"], m["distance"].astype(np.int64)], axis=1)
    mt = Matcher()
    args = (ix["cand_b_off"], ix["cand_b"], db, ix["cand_a_off"], ix["cand_a"])
    for case, (ka, kb) in {"plain": (None, None), "loose": (None, None), "masked": ("mask_a", "mask_b"), "nodiff": (None, "mask_b")}.items():
        md, mn = (int(v) for v in ix["par_" + case])
        m = mt.IndexedMatch(da, *args, md, mn, None if ka is None else ix[ka], None if kb is None else ix[kb])
        assert np.array_equal(as3(m), ix["exp_" + case]) and np.all(m["imgIdx"] == 0)
    rng = np.random.default_rng(21)
    for md, mn in ((10, 0), (45, 3), (255, 1), (0, 1)):
        ma = rng.random(len(da)) < 0.7; mb = rng.random(len(db)) < 0.7
        assert np.array_equal(mt.IndexedMatch(da, *args, md, mn, ma, mb), O.indexed_match(da, args[0], args[1], db, args[3], args[4], md, mn, ma, mb))
    assert len(mt.IndexedMatch(da, *args, 30, 1, np.zeros(len(da), bool), None)) == 0           # empty mask: FeatureMatcher.cpp:208
    assert len(mt.IndexedMatch(da[:0], [0], [], db, np.zeros(len(db) + 1, np.int32), [], 30, 1)) == 0
    bad = ix["cand_b"].copy(); bad[3] = len(db)
    with pytest.raises(MageError):
        mt.IndexedMatch(da, ix["cand_b_off"], bad, db, ix["cand_a_off"], ix["cand_a"], 30, 1)


def test_undistort_keypoints_golden_oracle_and_device_path(gold):
    """"next" row ORB-10: UndistortKeypoints.  Bit-exact against the fixture and the oracle; the device-resident form applied to
    the extractor's batch output equals the host form applied to the same keypoints; argument checks."""
    import os
    from mageslam_amd._lib import MageError
    from mageslam_amd.orb import UndistortParams
    u = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "orb_undistort.npz"))
    det = OrbDetector()
    k = np.zeros(len(u["xy"]), O.KEYPOINT_DTYPE)
    k["x"], k["y"], k["size"], k["response"], k["class_id"] = u["xy"][:, 0], u["xy"][:, 1], 15, 40, -1
    for model in ("poly3k", "rational6k"):
        par = UndistortParams.make(u["K"], u["dist_" + model], u["P"])
        g = det.UndistortKeypoints(k, par)
        assert np.array_equal(np.stack([g["x"], g["y"]], 1), u["exp_" + model])
        assert np.array_equal(g, O.undistort_keypoints(k, O.UndistortParams.make(u["K"], u["dist_" + model], u["P"])))
    assert len(det.UndistortKeypoints(k[:0], par)) == 0
    with pytest.raises(MageError):
        det.UndistortKeypoints(k, UndistortParams.make(u["K"], np.zeros(12), u["P"]))        # thin-prism terms: not in the reference's models
    with pytest.raises(MageError):
        det.UndistortKeypoints(k, UndistortParams.make(np.zeros(9), u["dist_poly3k"], u["P"]))
    # device-resident chain: detect a batch, undistort it in HBM, read the buffer back (plain HIP runtime calls through ctypes)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    cap = int(det.params.nfeatures)
    imgs = np.ascontiguousarray(np.stack([gold["orb_640x480_a_img"], gold["orb_640x480_b_img"]]))
    kp_host, _, cnt = det.DetectAndComputeBatch(imgs)
    exp = [det.UndistortKeypoints(kp_host[f, : cnt[f]], par) for f in range(2)]
    d_img = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d_img), imgs.nbytes) == 0
    assert hip.hipMemcpy(d_img, imgs.ctypes.data_as(ctypes.c_void_p), imgs.nbytes, 1) == 0                      # 1 = hipMemcpyHostToDevice
    kp_ptr, _, cn_ptr = det.detect_batch_device(d_img.value, 2, 640, 480)
    det.undistort_device(kp_ptr, cn_ptr, 2, cap, par)
    assert hip.hipDeviceSynchronize() == 0
    got = np.zeros((2, cap), O.KEYPOINT_DTYPE)
    assert hip.hipMemcpy(got.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(kp_ptr), got.nbytes, 2) == 0      # 2 = hipMemcpyDeviceToHost
    hip.hipFree(d_img)
    for f in range(2):
        assert np.array_equal(got[f, : cnt[f]], exp[f])


@pytest.mark.parametrize("name,patch", [("orb_160x120", 15), ("orb_640x480_a", 15), ("orb_160x120", 31)])
def test_oriented_detection_golden_and_oracle(gold, name, patch):
    """"next" row ORB-6: UseOrientation on the device -- keypoints, float32 angles and rotated-BRIEF descriptors bit-exact against
    the fixture (independent numpy) and the C oracle."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "orb_oriented.npz"))
    img = gold[name + "_img"]
    det = OrbDetector(default_params(use_orientation=1, patch_size=patch))
    k, d = det.DetectAndCompute(img)
    key = f"{name}_p{patch}"
    assert np.array_equal(np.stack([k["x"], k["y"], k["response"]], 1).astype(np.int64), g[key + "_kp"])
    assert np.array_equal(k["angle"], g[key + "_angle"])
    assert np.array_equal(d, g[key + "_desc"])
    ko, do = O.orb_detect(img, O.OrbParams.defaults(use_orientation=1, patch_size=patch))
    assert np.array_equal(k, ko) and np.array_equal(d, do)


@pytest.mark.parametrize("key", ["l3_160x120", "l2_s12_640x480", "l2_oriented_160x120", "l4_p31_160x120"])
def test_pyramid_detection_golden_and_oracle(gold, key):
    """NumLevels > 1 on the device: resize pyramid, per-level stages, concatenation -- bit-exact against the numpy fixture and the
    C oracle (float32 scaled coordinates included); a smaller capacity truncates like ImageData::Insert."""
    import os
    from test_orb_oracle import PYRAMID_CASES, check_pyramid_case
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "orb_pyramid.npz"))
    name, kw = PYRAMID_CASES[key]
    img = gold[name + "_img"]
    det = OrbDetector(default_params(**kw))
    k, d = det.DetectAndCompute(img)
    check_pyramid_case(k, d, g, key)
    ko, do = O.orb_detect(img, O.OrbParams.defaults(**kw))
    assert np.array_equal(k, ko) and np.array_equal(d, do)
    cap = len(ko) - 7
    k2, d2 = det.DetectAndCompute(img, capacity=cap)
    ko2, do2 = O.orb_detect(img, O.OrbParams.defaults(**kw), cap=cap)
    assert len(k2) == cap and np.array_equal(k2, ko2) and np.array_equal(d2, do2)
    # batch of two different frames of the same size: frames are independent
    if name == "orb_640x480_a":
        kb, db, cb = det.DetectAndComputeBatch(np.stack([gold["orb_640x480_b_img"], img]))
        assert cb[1] == len(ko) and np.array_equal(kb[1, : cb[1]], ko) and np.array_equal(db[1, : cb[1]], do)
        kob, dob = O.orb_detect(gold["orb_640x480_b_img"], O.OrbParams.defaults(**kw))
        assert cb[0] == len(kob) and np.array_equal(kb[0, : cb[0]], kob) and np.array_equal(db[0, : cb[0]], dob)


@pytest.mark.parametrize("patch", [21, 9])
def test_random_pattern_patch_sizes_golden_and_oracle(gold, patch):
    """ORB-9 on the device: the cv::RNG random pattern for patch sizes without a pre-rotated table (angle 0)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "orb_pyramid.npz"))
    img = gold["orb_160x120_img"]
    k, d = OrbDetector(default_params(patch_size=patch)).DetectAndCompute(img)
    assert np.array_equal(np.stack([k["x"], k["y"], k["response"]], 1).astype(np.int64), g[f"rand{patch}_kp"])
    assert np.array_equal(d, g[f"rand{patch}_desc"])
    ko, do = O.orb_detect(img, O.OrbParams.defaults(patch_size=patch))
    assert np.array_equal(k, ko) and np.array_equal(d, do)


@pytest.mark.parametrize("patch,name,extra", [(21, "orb_160x120", {}), (9, "orb_640x480_a", {}), (27, "orb_640x480_b", dict(nlevels=2, scale_factor=1.5)),
                                              (21, "orb_640x480_a", dict(nfeatures=1000, gaussian_kernel_size=5))])
def test_random_pattern_with_orientation_matches_oracle(gold, patch, name, extra):
    """ORB-9 with UseOrientation (ComputeOrbDescriptors, OpenCVModified.cpp:452-492) on the device: k_brief_rotated rotates the 512
    random points per keypoint (float cos / sin of the intensity-centroid angle, float rotation, cvRound).  Keypoints, float32
    angles and descriptors bit-exact against the C oracle (which tests/test_orb_oracle.py holds against the numpy twin)."""
    img = gold[name + "_img"]
    kw = dict(patch_size=patch, use_orientation=1, **extra)
    okw = {"feature_factor_anms": "feature_factor"}
    k, d = OrbDetector(default_params(**kw)).DetectAndCompute(img)
    ko, do = O.orb_detect(img, O.OrbParams.defaults(**{okw.get(a, a): b for a, b in kw.items()}), cap=max(440, kw.get("nfeatures", 440)))
    assert len(k) > 100
    assert np.array_equal(k, ko) and np.array_equal(d, do)


def test_cpp_front_end_shims_compile_and_agree_with_the_mirror(tmp_path):
    """include/OrbDetector.h + include/FeatureMatcher.h (the reference's OrbDetector / mage::OrbFeatureDetector classes and the
    Match / RadiusMatch / GetDescriptorDistance functions over the C ABI): tools/shim_orb_match.cpp, written against the two headers
    with stand-ins that have cv::KeyPoint / cv::DMatch / cv::Mat's layouts, builds with the host compiler alone and returns, record
    for record, what the Python mirror (itself checked against the oracle above) returns."""
    import os
    import shutil
    import subprocess
    from mageslam_amd.orb import UndistortParams
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        pytest.skip("no host C++ compiler")
    exe = str(tmp_path / "shim_orb_match")
    lib_dir = os.path.join(root, "mageslam_amd")
    subprocess.run([cxx, "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "tools", "shim_orb_match.cpp"),
                    "-L" + lib_dir, "-lmageslam_hip", "-Wl,-rpath," + lib_dir, "-o", exe], check=True, capture_output=True, timeout=300)
    a, b = frames.frame_pair(23)
    a.tofile(tmp_path / "a.raw"); b.tofile(tmp_path / "b.raw")
    r = subprocess.run([exe, str(tmp_path / "a.raw"), str(tmp_path / "b.raw"), "640", "480"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = [ln.split() for ln in r.stdout.splitlines()]
    det, mt = OrbDetector(), Matcher()
    ka, da = det.DetectAndCompute(a)
    kb, db = det.DetectAndCompute(b)
    K = np.array([500, 0, 320, 0, 500, 240, 0, 0, 1], np.float32); Kn = np.array([480, 0, 320, 0, 480, 240, 0, 0, 1], np.float32)
    kb = det.UndistortKeypoints(kb, UndistortParams.make(K.reshape(3, 3), np.array([0.08, -0.02, 0.001, -0.0005, 0.004], np.float32), Kn.reshape(3, 3)))
    assert rows[0] == ["A", str(len(ka)), "B", str(len(kb))]
    tag = lambda t: [x[1:] for x in rows if x[0] == t]
    sig = lambda d: int(d[0]) | (int(d[31]) << 8)
    for got, k, d in ((tag("ka"), ka, da), (tag("kb"), kb, db)):
        assert len(got) == len(k)
        for g, kk, dd in zip(got, k, d):
            assert np.float32(g[0]) == kk["x"] and np.float32(g[1]) == kk["y"] and float(g[2]) == kk["response"] and int(g[3]) == sig(dd)
    maskA = np.ones(len(ka), np.uint8); maskA[::5] = 0
    m = mt.Match(da, db, maskA, None, 40, 2)
    assert [[int(x[0]), int(x[1]), int(x[2]), float(x[3])] for x in tag("m")] == [[int(q["queryIdx"]), int(q["trainIdx"]), int(q["imgIdx"]), float(q["distance"])] for q in m]
    over = np.stack([ka["x"] + np.float32(1.5), ka["y"] - np.float32(0.5)], axis=1).astype(np.float32)
    rm = mt.RadiusMatch(ka, da, kb, db, 12.0, 50, 1, query_position_overrides=over)
    assert [[int(x[0]), int(x[1]), float(x[3])] for x in tag("r")] == [[int(q["queryIdx"]), int(q["trainIdx"]), float(q["distance"])] for q in rm]
    singles = []
    for i in range(min(20, len(ka))):
        one = mt.RadiusMatch(ka[i:i + 1], da[i:i + 1], kb, db, 12.0, 50, 1)
        if len(one):
            singles.append([i, int(one[0]["trainIdx"]), float(one[0]["distance"])])
    assert [[int(x[0]), int(x[1]), float(x[2])] for x in tag("s")] == singles
    assert tag("dist") == [[str(GetDescriptorDistance(da[0], db[0])), "0"]]


BOW_TREES = ("deep", "flat", "binary")


@pytest.mark.parametrize("tree", BOW_TREES)
def test_bow_leaf_lookup_golden_and_oracle(gold, tree):
    """SURVEY.md 8f rank 4, the vocabulary half of IndexedMatch: OnlineBow::FindLeafNode (BoW/OnlineBow.cpp:289-311) as a batch on the
    device, bit-exact against the numpy-made fixture (tied siblings, child lists out of node order, single-child nodes, a one-level
    tree) and against the C oracle on descriptors the fixture does not hold."""
    bw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "orb_bow.npz"))
    t = (bw[tree + "_nodes"], bw[tree + "_child_off"], bw[tree + "_children"])
    mt = Matcher()
    assert np.array_equal(mt.BowFindLeaf(*t, gold["orb_640x480_a_desc"]), bw[tree + "_leaf_a"])
    assert np.array_equal(mt.BowFindLeaf(*t, gold["orb_640x480_b_desc"]), bw[tree + "_leaf_b"])
    rng = np.random.default_rng(5)
    q = rng.integers(0, 256, (3000, 32), dtype=np.uint8)
    q[::7] = t[0][rng.integers(0, len(t[0]), len(q[::7]))]           # exact medoids: distance 0, ties with duplicated siblings
    assert np.array_equal(mt.BowFindLeaf(*t, q), O.bow_find_leaf(*t, q))
    assert len(mt.BowFindLeaf(*t, q[:0])) == 0
    assert mt.BowFindLeaf(t[0][:1], [0, 0], [], q[:5]).tolist() == [0] * 5           # the root alone is its own leaf


def test_resident_vocabulary_gives_the_same_leaves_and_matches(gold):
    """mage_bow_set_tree: lookups and IndexedMatch with the tree kept on the device equal the ones that stage it per call (and the
    fixture); without any tree the call is refused."""
    bw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "orb_bow.npz"))
    da, db = gold["orb_640x480_a_desc"], gold["orb_640x480_b_desc"]
    mt = Matcher()
    with pytest.raises(Exception):
        mt.BowFindLeaf(None, None, None, da)
    for tree in BOW_TREES:
        t = (bw[tree + "_nodes"], bw[tree + "_child_off"], bw[tree + "_children"])
        mt.BowSetTree(*t)
        la, lb = mt.BowFindLeaf(None, None, None, da), mt.BowFindLeaf(None, None, None, db)
        assert np.array_equal(la, bw[tree + "_leaf_a"]) and np.array_equal(lb, bw[tree + "_leaf_b"])

        def csr(leaves):
            order = np.argsort(leaves, kind="stable").astype(np.int32)
            off = np.zeros(len(t[0]) + 1, np.int32)
            np.add.at(off, leaves + 1, 1)
            return np.cumsum(off).astype(np.int32), order
        fao, fa = csr(la); fbo, fb = csr(lb)
        kept = mt.IndexedMatchBow(None, None, None, da, fao, fa, db, fbo, fb, 40, 2)
        staged = mt.IndexedMatchBow(*t, da, fao, fa, db, fbo, fb, 40, 2)
        assert len(kept) > 0 and np.array_equal(kept.view(np.uint8), staged.view(np.uint8))
    mt.BowSetTree()
    with pytest.raises(Exception):
        mt.BowFindLeaf(None, None, None, da)


@pytest.mark.parametrize("tree", BOW_TREES)
def test_indexed_match_through_the_vocabulary(gold, tree):
    """mage_match_indexed_bow: IndexedMatch with its candidate lists looked up in the tree on the device (FeatureMatcher.cpp:223-227,
    253-257 -> OnlineBow::QueryFeatures): the fixture's matches, the oracle's on other limits, and the same records as
    mage_match_indexed fed with the lists a host-side lookup builds."""
    bw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "orb_bow.npz"))
    da, db = gold["orb_640x480_a_desc"], gold["orb_640x480_b_desc"]
    t = (bw[tree + "_nodes"], bw[tree + "_child_off"], bw[tree + "_children"])
    fa, fb = (bw[tree + "_feat_a_off"], bw[tree + "_feat_a"]), (bw[tree + "_feat_b_off"], bw[tree + "_feat_b"])
    mt = Matcher()
    for case in ("plain", "loose", "masked"):
        md, mn = (int(v) for v in bw[f"{tree}_par_{case}"])
        ma, mb = (bw["mask_a"], bw["mask_b"]) if case == "masked" else (None, None)
        m = mt.IndexedMatchBow(*t, da, *fa, db, *fb, md, mn, ma, mb)
        got = np.stack([m["queryIdx"], m["trainIdx"], m["distance"].astype(np.int64)], axis=1)
        assert np.array_equal(got, bw[f"{tree}_exp_{case}"]) and np.all(m["imgIdx"] == 0), case
    for md, mn in ((12, 0), (256, 1), (40, 9)):
        assert np.array_equal(mt.IndexedMatchBow(*t, da, *fa, db, *fb, md, mn, bw["mask_a"], None), O.indexed_match_bow(*t, da, *fa, db, *fb, md, mn, bw["mask_a"], None))
    # the same through explicit candidate lists
    la, lb = bw[tree + "_leaf_a"], bw[tree + "_leaf_b"]
    cb = [fb[1][fb[0][l]:fb[0][l + 1]] for l in la]; ca = [fa[1][fa[0][l]:fa[0][l + 1]] for l in lb]
    csr = lambda ls: (np.concatenate([[0], np.cumsum([len(x) for x in ls])]).astype(np.int32), np.concatenate(ls + [np.zeros(0, np.int32)]).astype(np.int32))
    assert np.array_equal(mt.IndexedMatchBow(*t, da, *fa, db, *fb, 30, 1), mt.IndexedMatch(da, *csr(cb), db, *csr(ca), 30, 1))
    assert len(mt.IndexedMatchBow(*t, da, *fa, db, *fb, 30, 1, np.zeros(len(da), bool), None)) == 0
    with pytest.raises(Exception):          # a child that is not behind its parent would let a walk loop
        mt.BowFindLeaf(t[0][:2], [0, 1, 2], [1, 0], da[:4])
    with pytest.raises(Exception):          # a filed feature outside the image
        mt.IndexedMatchBow(*t, da, fa[0], np.where(fa[1] == 0, len(da), fa[1]), db, *fb, 30, 1)
