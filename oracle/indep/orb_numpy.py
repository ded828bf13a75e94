"""Independent numpy restatement of the ORB front-end and the brute-force matcher -- TEST INFRASTRUCTURE ONLY.

Pins oracle/orb_oracle.c and oracle/match_oracle.c.  Different routes on purpose:
  * FAST from its definition: "9 contiguous ring pixels all brighter than I+t or all darker than I-t", with the
    score as max over the 16 arcs of the arc's minimum margin (no threshold table, no min/max ladder);
  * NMS with shifted arrays; selection with numpy masks / lexsort; ANMS radii by the same grid-ring search but
    over Python dicts (the ring search is part of the behaviour being pinned: brute force is NOT equivalent when the
    bounding box is smaller than the grid);
  * blur with scipy.ndimage.correlate1d(mode="mirror"); BRIEF as one fancy-indexing gather;
  * Hamming through np.unpackbits.
"""
from __future__ import annotations

import numpy as np
from scipy.ndimage import correlate1d

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]

DEFAULTS = dict(gaussian_kernel_size=7, nfeatures=440, scale_factor=1.5, nlevels=1, patch_size=15, fast_threshold=4, feature_factor=1.5,
                feature_strength=0.9, strong_response=20, min_robust=1.1, max_robust=2.0, cells_x=32, cells_y=32)


def fast_score_map(img: np.ndarray, t: int) -> np.ndarray:
    img = img.astype(np.int32)
    h, w = img.shape
    score = np.zeros((h, w), np.uint8)
    if h < 7 or w < 7:
        return score
    c = img[3:h - 3, 3:w - 3]
    d = np.stack([c - img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in RING])          # centre minus ring
    d2 = np.concatenate([d, d[:8]])                                                           # wrap for arcs of 9
    arc_min = np.stack([d2[k:k + 9].min(axis=0) for k in range(16)]).max(axis=0)              # darker-ring margin
    arc_max = np.stack([(-d2[k:k + 9]).min(axis=0) for k in range(16)]).max(axis=0)           # brighter-ring margin
    m = np.maximum(arc_min, arc_max)
    corner = m > t
    score[3:h - 3, 3:w - 3] = np.where(corner, m - 1, 0).astype(np.uint8)
    return score


def fast_keypoints(img: np.ndarray, t: int) -> np.ndarray:
    s = fast_score_map(img, t).astype(np.int32)
    h, w = s.shape
    p = np.pad(s, 1)
    nb = [p[1 + dy:1 + dy + h, 1 + dx:1 + dx + w] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy, dx) != (0, 0)]
    keep = np.ones_like(s, bool)
    for n in nb:
        keep &= s > n
    keep[:3] = False; keep[h - 3:] = False; keep[:, :3] = False; keep[:, w - 3:] = False
    ys, xs = np.nonzero(keep)                      # raster order
    return np.stack([xs, ys, s[ys, xs]], axis=1).astype(np.int32)


def retain_best(k: np.ndarray, min_thr: int, max_num: int, min_num: int, factor: float) -> np.ndarray:
    hist = np.bincount(np.clip(k[:, 2], 0, 255), minlength=256)
    cum = np.cumsum(hist[::-1])[::-1]              # cum[i] = #{resp >= i}
    mnt = min_thr
    for i in range(255, min_thr - 1, -1):
        if cum[i] >= min_num:
            mnt = i
            break
    lower = max(int(np.float32(mnt) * np.float32(factor)), min_thr)
    cut = lower
    for i in range(255, lower - 1, -1):
        if cum[i] >= max_num:
            cut = i
            break
    return k[k[:, 2] >= cut]


def anms(k: np.ndarray, keep: int, thr: int, P: dict) -> np.ndarray:
    n = len(k)
    if keep > n:
        return k
    x, y, s = k[:, 0].astype(int), k[:, 1].astype(int), k[:, 2].astype(np.float32)
    minX, maxX, minY, maxY = x.min(), x.max(), y.min(), y.max()
    nx, ny = P["cells_x"], P["cells_y"]
    hi = np.float32(P["strong_response"]) - np.float32(thr)
    val = np.clip(np.float32(s.min()) - np.float32(thr), np.float32(0), hi)
    rng = max(np.float32(0), np.float32(P["max_robust"]) - np.float32(P["min_robust"]))
    rf = np.float32(P["max_robust"]) - (val / np.float32(P["strong_response"] - thr)) * rng
    cells: dict = {}
    cxs = (x - minX) * nx // (maxX + 1 - minX); cys = (y - minY) * ny // (maxY + 1 - minY)
    for i in range(n):
        cells.setdefault((cxs[i], cys[i]), []).append(i)
    gmax = int(float(maxX - minX) * float(maxY - minY) / float(keep))
    md = min(max((maxX - minX) // nx, 1), max((maxY - minY) // ny, 1)) ** 2
    r = np.zeros(n, np.int64)
    for i in range(n):
        si = np.float32(s[i] * rf + np.float32(0.002))
        minr = gmax
        d = 0
        while max(0, d - 1) ** 2 * md < minr:
            for yy in range(-d, d + 1):
                for xx in range(-d, d + 1):
                    if max(abs(xx), abs(yy)) != d:
                        continue
                    for j in cells.get((cxs[i] + xx, cys[i] + yy), ()):
                        if s[j] > si:
                            minr = min(minr, (x[i] - x[j]) ** 2 + (y[i] - y[j]) ** 2)
            d += 1
        r[i] = minr
    order = np.lexsort((np.arange(n), -s, -r))     # r desc, strength desc, idx asc
    return k[order[:keep]]


def gaussian_taps(ksize: int) -> np.ndarray:
    xs = np.arange(ksize) - (ksize - 1) * 0.5
    cf = np.exp(-0.5 / 4.0 * xs * xs).astype(np.float32)
    inv = 1.0 / float(np.sum(cf.astype(np.float64)))
    cf = (cf.astype(np.float64) * inv).astype(np.float32)
    return np.rint(cf.astype(np.float64) * 256.0).astype(np.int64)


def blur(img: np.ndarray, ksize: int) -> np.ndarray:
    t = gaussian_taps(ksize)
    a = correlate1d(img.astype(np.int64), t, axis=1, mode="mirror")
    b = correlate1d(a, t, axis=0, mode="mirror")
    return np.minimum((b + 32768) >> 16, 255).astype(np.uint8)


def random_pattern(patch: int) -> np.ndarray:
    """MakeRandomPattern (OpenCVModified.cpp:551-560): cv::RNG(0x34985739), uniform(-patch/2, patch/2+1) for x then y of 512 points.
    Python integers, so the 64-bit multiply-with-carry state needs explicit masks.  Returns the 1024 coordinates."""
    state = 0x34985739
    lo, hi = -(patch // 2), patch // 2 + 1          # C++ -patchSize / 2 truncates towards zero: equal to -(patch // 2) for patch > 0
    out = np.zeros(1024, np.int64)
    for i in range(1024):
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
        out[i] = lo + (state & 0xFFFFFFFF) % (hi - lo)
    return out


def expand_pattern(base: np.ndarray) -> np.ndarray:
    out = np.zeros((30, 512, 2), np.int64)
    b = base.reshape(512, 2).astype(np.float64)
    for k in range(30):
        a = np.deg2rad(12.0 * k)
        v = np.stack([b[:, 0] * np.cos(a) - b[:, 1] * np.sin(a), b[:, 0] * np.sin(a) + b[:, 1] * np.cos(a)], axis=1)
        hv = np.rint(v * 2) / 2
        v = np.where(np.abs(v - hv) < 1e-9, hv, v)
        out[k] = np.rint(v)
    return out


def fast_atan2_deg(y, x):
    """cv::fastAtan2 (OpenCV 3.4.0): 7th-order odd polynomial of min/max ratio, float32 throughout, degrees in [0, 360]."""
    f = np.float32
    y = np.asarray(y, f); x = np.asarray(x, f)
    scale = f(180 / 3.1415926535897932384626433832795)
    p1, p3, p5, p7 = f(0.9997878412794807) * scale, f(-0.3258083974640975) * scale, f(0.1555786518463281) * scale, f(-0.04432655554792128) * scale
    ax, ay = np.abs(x), np.abs(y)
    eps = f(np.finfo(np.float64).eps)
    swap = ax < ay
    num = np.where(swap, ax, ay); den = np.where(swap, ay, ax) + eps
    c = (num / den).astype(f); c2 = (c * c).astype(f)
    a = ((((p7 * c2 + p5).astype(f) * c2 + p3).astype(f) * c2 + p1).astype(f) * c).astype(f)
    a = np.where(swap, f(90) - a, a).astype(f)
    a = np.where(x < 0, f(180) - a, a).astype(f)
    a = np.where(y < 0, f(360) - a, a).astype(f)
    return a


def ic_angles(img: np.ndarray, k: np.ndarray, half: int) -> np.ndarray:
    """Intensity-centroid orientation (OpenCVModified.cpp:399-437): first moments over the discretised disc of radius `half`
    (row v spans |u| <= umax[v], the table of :672-688), angle = fastAtan2(m01, m10)."""
    vmax = int(np.floor(half * np.sqrt(np.float32(2)) / 2 + 1)); vmin = int(np.ceil(half * np.sqrt(np.float32(2)) / 2))
    umax = np.zeros(half + 2, np.int64)
    for v in range(vmax + 1):
        umax[v] = int(np.rint(np.sqrt(float(half * half - v * v))))
    v0 = 0
    for v in range(half, vmin - 1, -1):
        while umax[v0] == umax[v0 + 1]:
            v0 += 1
        umax[v] = v0
        v0 += 1
    vv, uu = np.mgrid[-half:half + 1, -half:half + 1]
    disc = np.abs(uu) <= umax[np.abs(vv)]
    I = img.astype(np.int64)
    out = np.zeros(len(k), np.float32)
    for i, (x, y) in enumerate(k[:, :2]):
        patch = I[y - half:y + half + 1, x - half:x + half + 1] * disc
        out[i] = fast_atan2_deg(np.float32((patch * vv).sum()), np.float32((patch * uu).sum()))
    return out


def resize_linear(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv::resize(..., INTER_LINEAR) on CV_8UC1 as OpenCV 3.4.0 computes it: 11-bit fixed-point weights (rounded half to even, as
    saturate_cast<short>), horizontal pass in int, vertical pass ((b*(S>>4))>>16 twice, +2)>>2.  Vectorised."""
    sh, sw = src.shape
    f = np.float32
    fx = ((np.arange(dw) + 0.5) * (sw / dw) - 0.5).astype(f)
    sx = np.floor(fx).astype(np.int64); fx = (fx - sx.astype(f)).astype(f)
    fx = np.where(sx < 0, f(0), fx); sx = np.maximum(sx, 0)
    fx = np.where(sx >= sw - 1, f(0), fx); sx = np.minimum(sx, sw - 1)
    a0 = np.clip(np.rint((f(1) - fx) * f(2048)), -32768, 32767).astype(np.int64); a1 = np.clip(np.rint(fx * f(2048)), -32768, 32767).astype(np.int64)
    fy = ((np.arange(dh) + 0.5) * (sh / dh) - 0.5).astype(f)
    sy = np.floor(fy).astype(np.int64); fy = (fy - sy.astype(f)).astype(f)
    b0 = np.clip(np.rint((f(1) - fy) * f(2048)), -32768, 32767).astype(np.int64); b1 = np.clip(np.rint(fy * f(2048)), -32768, 32767).astype(np.int64)
    r0 = np.clip(sy, 0, sh - 1); r1 = np.clip(sy + 1, 0, sh - 1)
    S = src.astype(np.int64)
    H = S[:, sx] * a0[None, :] + S[:, np.minimum(sx + 1, sw - 1)] * a1[None, :]          # (sh, dw)
    out = (((b0[:, None] * (H[r0] >> 4)) >> 16) + ((b1[:, None] * (H[r1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def level_layout(w: int, h: int, P: dict):
    """Scales, sizes and per-level feature quotas of the pyramid (OpenCVModified.cpp:564-567, :797-799, :659-669)."""
    f = np.float32
    L = P["nlevels"]
    scales = [f(np.float64(P["scale_factor"]) ** l) for l in range(L)]
    sizes = [(int(np.rint(f(w) / s)), int(np.rint(f(h) / s))) for s in scales]
    factor = f(1) / f(P["scale_factor"])
    nd = f(P["nfeatures"]) * (f(1) - factor) / (f(1) - f(np.float64(factor) ** L))
    quota, tot = [], 0
    for _ in range(L - 1):
        quota.append(int(np.rint(nd))); tot += quota[-1]; nd = f(nd * factor)
    quota.append(max(P["nfeatures"] - tot, 0))
    if L == 1:
        quota = [P["nfeatures"]]
    return scales, sizes, quota


def detect(img: np.ndarray, base_pattern: np.ndarray, **kw):
    """Returns (keypoints x y response [level coordinates], descriptors, blurred level 0[, angles when use_orientation]); with
    nlevels > 1 the keypoint rows carry a fourth column, the octave, and `scaled_xy` gives the float32 image coordinates."""
    P = dict(DEFAULTS); P.update(kw)
    h, w = img.shape
    scales, sizes, quota = level_layout(w, h, P)
    b = P["patch_size"] // 2
    half_patch = b
    if P.get("use_orientation"):
        b = int(np.ceil(np.float32(b) * np.sqrt(np.float32(2))))
    levels = [img]
    for l in range(1, P["nlevels"]):
        levels.append(resize_linear(levels[-1], sizes[l][0], sizes[l][1]))
    ks, octs = [], []
    for l, im in enumerate(levels):
        hh, ww = im.shape
        if hh < 7 or ww < 7 or quota[l] < 1:
            continue
        k = fast_keypoints(im, P["fast_threshold"])
        if hh <= 2 * b or ww <= 2 * b:
            k = k[:0]
        else:
            k = k[(k[:, 0] >= b) & (k[:, 0] < ww - b) & (k[:, 1] >= b) & (k[:, 1] < hh - b)]
        if len(k) > quota[l]:
            k = retain_best(k, P["fast_threshold"], int(np.float32(quota[l]) * np.float32(P["feature_factor"])), quota[l], P["feature_strength"])
            k = anms(k, quota[l], P["fast_threshold"], P)
        ks.append(k); octs.append(np.full(len(k), l, np.int64))
    k = np.concatenate(ks) if ks else np.zeros((0, 3), np.int64)
    octave = np.concatenate(octs) if octs else np.zeros(0, np.int64)
    blurred = [blur(im, P["gaussian_kernel_size"]) if P["gaussian_kernel_size"] > 1 else im for im in levels]
    if P["patch_size"] in (15, 31):
        table = expand_pattern(base_pattern).reshape(30, 256, 4)
    else:                                           # random pattern: the unrotated points; with orientation every keypoint rotates them itself
        table = np.zeros((30, 256, 4), np.int64); table[0] = random_pattern(P["patch_size"]).reshape(256, 4)
    if P.get("use_orientation"):
        ang = np.zeros(len(k), np.float32)
        for l in range(P["nlevels"]):
            m = octave == l
            if m.any():
                ang[m] = ic_angles(levels[l], k[m], half_patch)
        inc = np.rint(ang / np.float32(12)).astype(np.int64) % 30          # cvRound: half to even, like np.rint
    else:
        ang = None
        inc = np.zeros(len(k), np.int64)
    desc = np.zeros((len(k), 32), np.uint8)
    for l in range(P["nlevels"]):
        m = np.nonzero(octave == l)[0]
        if not len(m):
            continue
        if P.get("use_orientation") and P["patch_size"] not in (15, 31):
            # ComputeOrbDescriptors (OpenCVModified.cpp:452-492): float angle in radians, cos / sin rounded to float, points rotated in
            # float32 and rounded half-to-even (cvRound)
            f = np.float32
            rad = ang[m] * f(np.pi / 180.0)
            a = np.cos(rad.astype(np.float64)).astype(f)[:, None]; b_ = np.sin(rad.astype(np.float64)).astype(f)[:, None]
            base = table[0].astype(f)                                             # (256, 4): x0 y0 x1 y1
            def rot(px, py):
                return np.rint(px[None, :] * a - py[None, :] * b_).astype(np.int64), np.rint(px[None, :] * b_ + py[None, :] * a).astype(np.int64)
            x0, y0 = rot(base[:, 0], base[:, 1]); x1, y1 = rot(base[:, 2], base[:, 3])
            pat = np.stack([x0, y0, x1, y1], axis=2)                               # (n, 256, 4)
        else:
            pat = table[inc[m]]                                                   # (n, 256, 4)
        xs, ys = k[m, 0][:, None], k[m, 1][:, None]
        bl = blurred[l]
        t0 = bl[ys + pat[:, :, 1], xs + pat[:, :, 0]]
        t1 = bl[ys + pat[:, :, 3], xs + pat[:, :, 2]]
        bits = (t0 < t1).astype(np.uint8).reshape(len(m), 32, 8)
        desc[m] = np.packbits(bits, axis=2, bitorder="little").reshape(len(m), 32)
    if P["nlevels"] > 1:
        k = np.concatenate([k, octave[:, None]], axis=1)
    return (k, desc, blurred[0]) if ang is None else (k, desc, blurred[0], ang)


def scaled_xy(k: np.ndarray, P: dict) -> np.ndarray:
    """float32 image coordinates of multi-level keypoints: pt *= scale of the octave (OpenCVModified.cpp:756-760)."""
    f = np.float32
    sc = np.array([f(np.float64(P["scale_factor"]) ** int(o)) for o in k[:, 3]], f)
    return np.stack([k[:, 0].astype(f) * sc, k[:, 1].astype(f) * sc], 1)


def hamming_matrix(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    a = np.unpackbits(A.reshape(-1, 32), axis=1).astype(np.int32)
    b = np.unpackbits(B.reshape(-1, 32), axis=1).astype(np.int32)
    return a @ (1 - b).T + (1 - a) @ b.T


def match(A: np.ndarray, B: np.ndarray, max_dist=30, min_diff=1):
    if len(A) == 0 or len(B) == 0:
        return np.zeros((0, 3), np.int64)
    D = hamming_matrix(A, B)

    def oneway(D):
        best = np.full(D.shape[0], -1)
        dist = np.zeros(D.shape[0], np.int64)
        for q in range(D.shape[0]):
            cand = np.nonzero(D[q] <= max_dist)[0]
            if cand.size == 0:
                continue
            order = cand[np.argsort(D[q, cand], kind="stable")]
            if cand.size > 1 and D[q, order[1]] - D[q, order[0]] < min_diff:
                continue
            best[q] = order[0]; dist[q] = D[q, order[0]]
        return best, dist
    f, fd = oneway(D)
    g, _ = oneway(D.T)
    out = [(q, f[q], fd[q]) for q in range(len(A)) if f[q] >= 0 and g[f[q]] == q]
    return np.array(out, np.int64).reshape(-1, 3)


def radius_match(qxy, qoct, qdesc, txy, toct, tdesc, radius, max_dist=30, min_diff=1, qmask=None, tmask=None):
    """Independent restatement of RadiusMatch (FeatureMatcher.cpp:294-446) with candidates in ascending target index."""
    qxy = np.asarray(qxy, np.float32); txy = np.asarray(txy, np.float32); r = np.float32(radius)
    D = hamming_matrix(qdesc, tdesc) if len(qdesc) and len(tdesc) else np.zeros((len(qdesc), len(tdesc)), np.int64)
    almost = []
    for q in range(len(qxy)):
        if qmask is not None and not qmask[q]:
            continue
        inside = ((txy[:, 0] >= qxy[q, 0] - r) & (txy[:, 0] <= qxy[q, 0] + r) & (txy[:, 1] >= qxy[q, 1] - r) & (txy[:, 1] <= qxy[q, 1] + r)
                  & (np.abs(np.asarray(toct) - qoct[q]) * 100 <= 1))
        if tmask is not None:
            inside &= np.asarray(tmask, bool)
        best, second, train = max_dist + 1, 2 ** 31 - 1, -1
        for t in np.nonzero(inside)[0]:
            if D[q, t] < best:
                train, second, best = t, best, int(D[q, t])
        if train >= 0 and second - best > min_diff:
            almost.append((q, train, best))
    if len(almost) > 1:
        by_t = {}
        for q, t, d in almost:
            by_t.setdefault(t, []).append(d)
        keep = []
        for q, t, d in almost:
            ds = sorted(by_t[t])
            if d == ds[0] and (len(ds) == 1 or ds[0] < ds[1]):
                keep.append((q, t, d))
        almost = keep
    return np.array(almost, np.int64).reshape(-1, 3)


def indexed_match(descA, cand_b, descB, cand_a, max_dist=30, min_diff=1, maskA=None, maskB=None):
    """Independent restatement of IndexedMatch (FeatureMatcher.cpp:192-292): cand_b[a] / cand_a[b] are python lists of candidate
    indices in the order the vocabulary index returned them.  Returns rows (queryIdx, trainIdx, distance)."""
    nA, nB = len(descA), len(descB)
    mA = np.ones(nA, bool) if maskA is None else np.asarray(maskA, bool)
    mB = np.ones(nB, bool) if maskB is None else np.asarray(maskB, bool)
    if not mA.any() or not mB.any():
        return np.zeros((0, 3), np.int64)
    D = hamming_matrix(descA, descB)
    limit = max_dist + 1

    def two_best(dist_of, cands, mask):
        ranked = []                      # (distance, arrival order, index): strict '<' updates = stable order on distance
        for order, c in enumerate(cands):
            if mask[c] and dist_of(c) < limit:
                ranked.append((int(dist_of(c)), order, int(c)))
        ranked.sort()
        best = ranked[0] if ranked else None
        second = ranked[1] if len(ranked) > 1 else None
        return best, second

    def passes(best, second):
        return best is not None and (second is None or second[0] - best[0] >= min_diff)

    out = []
    for a in range(nA):
        if not mA[a]:
            continue
        best, second = two_best(lambda b: D[a, b], cand_b[a], mB)
        if not passes(best, second):
            continue
        b = best[2]
        rbest, rsecond = two_best(lambda a2: D[a2, b], cand_a[b], mA)
        if passes(rbest, rsecond) and rbest[2] == a:
            out.append((a, b, rbest[0]))
    return np.array(out, np.int64).reshape(-1, 3)


def undistort_points(xy, K, dist, P, iterations=5):
    """Independent restatement of cv::undistortPoints (OpenCV 3.4.0) for float32 points: vectorised numpy, homogeneous matrices for
    the normalisation and the re-projection, radial terms by np.polyval -- same mathematics, different arithmetic route."""
    xy = np.asarray(xy, np.float32).astype(np.float64)
    K = np.asarray(K, np.float32).astype(np.float64).reshape(3, 3); P = np.asarray(P, np.float32).astype(np.float64).reshape(3, 3)
    d = np.zeros(8); dd = np.asarray(dist, np.float32).astype(np.float64).reshape(-1); d[:len(dd)] = dd
    k1, k2, p1, p2, k3, k4, k5, k6 = d
    x0 = (xy[:, 0] - K[0, 2]) / K[0, 0]
    y0 = (xy[:, 1] - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    for _ in range(iterations):
        r2 = x * x + y * y
        num = np.polyval([k6, k5, k4, 1.0], r2)
        den = np.polyval([k3, k2, k1, 1.0], r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (x0 - dx) * num / den
        y = (y0 - dy) * num / den
    h = P @ np.stack([x, y, np.ones_like(x)])
    return (h[:2] / h[2]).T.astype(np.float32)


def bow_find_leaf(node_desc, children_of, queries):
    """Independent restatement of OnlineBow::FindLeafNode (BoW/OnlineBow.cpp:289-311): children_of[n] is node n's child list (python
    lists); all queries walk the tree level by level, np.argmin picks the FIRST nearest child (the reference's strict '<')."""
    queries = np.asarray(queries, np.uint8).reshape(-1, 32)
    cur = np.zeros(len(queries), np.int64)
    active = np.array([len(children_of[0]) > 0] * len(queries), bool)
    while active.any():
        for n in np.unique(cur[active]):
            sel = active & (cur == n)
            kids = np.asarray(children_of[int(n)], np.int64)
            D = hamming_matrix(queries[sel], np.asarray(node_desc, np.uint8)[kids])
            cur[sel] = kids[np.argmin(D, axis=1)]
        active = np.array([len(children_of[int(c)]) > 0 for c in cur], bool)
    return cur


def indexed_match_bow(node_desc, children_of, descA, feat_a_of_leaf, descB, feat_b_of_leaf, max_dist=30, min_diff=1, maskA=None, maskB=None):
    """IndexedMatch with the candidate lists the vocabulary gives (FeatureMatcher.cpp:223-227, 253-257; OnlineBow::QueryFeatures :115-132):
    feat_x_of_leaf[n] = indices of image x's features filed under node n, in filing order."""
    la, lb = bow_find_leaf(node_desc, children_of, descA), bow_find_leaf(node_desc, children_of, descB)
    cand_b = [list(feat_b_of_leaf[int(l)]) for l in la]
    cand_a = [list(feat_a_of_leaf[int(l)]) for l in lb]
    return indexed_match(descA, cand_b, descB, cand_a, max_dist, min_diff, maskA, maskB)
