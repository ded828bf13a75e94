"""Deterministic synthetic bundle-adjustment scenes.

Host-side utility used by tests and bench.py to feed the BundlerLib surface with the shapes
BASELINE.json names (10/200/2k, 20/5k/50k, 1k/100k/1M ...).  Everything is produced by a
counter-based SplitMix64 hash so any element can be generated independently and vectorised:

    u64(seed, stream, index, k) = mix64(seed ^ (stream * C1) + index * C2 + (k + 1) * GAMMA)

with the SplitMix64 finaliser as ``mix64``; uniform double = (u64 >> 11) * 2**-53; normals by
Box-Muller (cos branch) on two consecutive draws.  The interface handed to BundlerLib is float32,
as the reference's is (BundlerLib.h:28-39), so every array below is cast to f32 at the end.

Conventions (reference: Core/MAGESLAM/Source/Data/Pose.cpp:63-87, Data/Intrinsics.h:25-46):
poses are world->camera (R, t = -R C); intrinsics are (cx, cy, fx, fy); the bundler projects with
fx for both axes (BundlerLib.cpp:266).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xD1342543DE82EF95)
_C2 = np.uint64(0xA0761D6478BD642F)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

# named streams
_S_YAW, _S_PT, _S_NOISE, _S_CAMPERT, _S_PTPERT, _S_OUTLIER = (np.uint64(i) for i in range(1, 7))


def _mix64(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def u64(seed: int, stream, index, k) -> np.ndarray:
    with np.errstate(over="ignore"):
        index = np.asarray(index, dtype=np.uint64)
        k = np.asarray(k, dtype=np.uint64)
        z = (np.uint64(seed) ^ (np.uint64(stream) * _C1)) + index * _C2 + (k + np.uint64(1)) * _GAMMA
        return _mix64(z)


def uniform(seed, stream, index, k) -> np.ndarray:
    return (u64(seed, stream, index, k) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def normal(seed, stream, index, k) -> np.ndarray:
    """k-th normal of (stream, index): Box-Muller cos branch on draws 2k, 2k+1."""
    k = np.asarray(k, dtype=np.uint64)
    u1 = 1.0 - uniform(seed, stream, index, 2 * k)          # (0, 1]
    u2 = uniform(seed, stream, index, 2 * k + np.uint64(1))
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def _rot_y(theta: np.ndarray) -> np.ndarray:
    c, s = np.cos(theta), np.sin(theta)
    R = np.zeros(theta.shape + (3, 3))
    R[..., 0, 0] = c; R[..., 0, 2] = s; R[..., 1, 1] = 1.0; R[..., 2, 0] = -s; R[..., 2, 2] = c
    return R


def so3_exp(w: np.ndarray) -> np.ndarray:
    """Rodrigues, vectorised over leading dims."""
    th = np.linalg.norm(w, axis=-1)
    K = np.zeros(w.shape[:-1] + (3, 3))
    K[..., 0, 1] = -w[..., 2]; K[..., 0, 2] = w[..., 1]
    K[..., 1, 0] = w[..., 2]; K[..., 1, 2] = -w[..., 0]
    K[..., 2, 0] = -w[..., 1]; K[..., 2, 1] = w[..., 0]
    th2 = np.where(th < 1e-8, 1.0, th)
    a = np.where(th < 1e-8, 1.0, np.sin(th) / th2)[..., None, None]
    b = np.where(th < 1e-8, 0.5, (1 - np.cos(th)) / (th2 * th2))[..., None, None]
    return np.eye(3) + a * K + b * (K @ K)


@dataclass
class Scene:
    """One BA problem in the float32 form the BundlerLib surface takes."""
    n_cams: int
    n_pts: int
    n_obs: int
    cam_t: np.ndarray          # (n_cams, 3) f32  world->camera translation (initial, perturbed)
    cam_R: np.ndarray          # (n_cams, 3, 3) f32 world->camera rotation, ROW-major here
    cam_K: np.ndarray          # (n_cams, 4) f32  cx, cy, fx, fy
    cam_fixed: np.ndarray      # (n_cams,) bool
    points: np.ndarray         # (n_pts, 3) f32 (initial, perturbed)
    obs_uv: np.ndarray         # (n_obs, 2) f32
    obs_cam: np.ndarray        # (n_obs,) u32
    obs_pt: np.ndarray         # (n_obs,) u32
    obs_info: np.ndarray       # (n_obs,) f32
    gt_cam_t: np.ndarray = field(default=None)
    gt_cam_R: np.ndarray = field(default=None)
    gt_points: np.ndarray = field(default=None)
    outlier_mask: np.ndarray = field(default=None)
    tethers: "Tethers" = field(default=None)      # optional pose-pose constraints (stereo rigs, sensor fusion)

    def cam_R_colmajor(self) -> np.ndarray:
        """(n_cams, 9) f32, column-major 3x3 as Eigen::Map<const Matrix3f> expects."""
        return np.ascontiguousarray(self.cam_R.transpose(0, 2, 1).reshape(self.n_cams, 9))


@dataclass
class Tethers:
    """Pose-pose constraints in the float32 form of BundlerLib.h:40-47 (quaternions x, y, z, w as Eigen stores them).

    distance : (d - |t_b - t_a|) * w                       BundlerLib.cpp:45-51
    rotation : angle((T_a^-1 T_b).rotation, q) * w          BundlerLib.cpp:76-87
    transform: log(T_b^-1 * SE3(q, p) * T_a), info w * I6   g2o EdgeSE3Expmap, BundlerLib.cpp:338-350
    """
    dist_cams: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.uint32))
    dist_d: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    dist_w: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    rot_cams: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.uint32))
    rot_q: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.float32))
    rot_w: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    xf_cams: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.uint32))
    xf_p: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    xf_q: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.float32))
    xf_w: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))


def quat_from_R(R: np.ndarray) -> np.ndarray:
    """(n, 3, 3) rotation matrices -> (n, 4) unit quaternions x, y, z, w with w >= 0 (Shepperd's method)."""
    R = np.asarray(R, np.float64).reshape(-1, 3, 3)
    q = np.zeros((R.shape[0], 4))
    for i, m in enumerate(R):
        c = np.array([m[0, 0], m[1, 1], m[2, 2], m[0, 0] + m[1, 1] + m[2, 2]])
        k = int(np.argmax(c))
        if k == 3:
            w = 0.5 * np.sqrt(1 + c[3])
            v = np.array([m[2, 1] - m[1, 2], m[0, 2] - m[2, 0], m[1, 0] - m[0, 1]]) / (4 * w)
            q[i] = [v[0], v[1], v[2], w]
        else:
            a, b, d = k, (k + 1) % 3, (k + 2) % 3
            s = 2 * np.sqrt(1 + m[a, a] - m[b, b] - m[d, d])
            q[i, a] = 0.25 * s
            q[i, b] = (m[b, a] + m[a, b]) / s
            q[i, d] = (m[d, a] + m[a, d]) / s
            q[i, 3] = (m[d, b] - m[b, d]) / s
        if q[i, 3] < 0:
            q[i] = -q[i]
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def make_tethers(scene: Scene, n_dist: int = 0, n_rot: int = 0, n_xf: int = 0, *, seed: int = 0x7E7E0001,
                 stride: int = 1, step: int = 3, weight: float = 30.0, noise: float = 1e-3) -> Tethers:
    """Tethers between cameras (i, i + stride) measured on the ground-truth poses plus a little noise, the way a
    stereo rig / inertial fuser would produce them (BundleAdjust.cpp:57-112).  Kinds cycle over the camera rail."""
    nc = scene.n_cams
    R, t = scene.gt_cam_R, scene.gt_cam_t
    S_T = np.uint64(9)

    def pairs(n, off):
        a = (off + np.arange(n) * step) % max(nc - stride, 1)
        return np.stack([a, a + stride], axis=1).astype(np.uint32)

    T = Tethers()
    if n_dist:
        c = pairs(n_dist, 0)
        d = np.linalg.norm(t[c[:, 1]] - t[c[:, 0]], axis=1) + noise * normal(seed, S_T, np.arange(n_dist), 0)
        T.dist_cams, T.dist_d, T.dist_w = c, d.astype(np.float32), np.full(n_dist, weight, np.float32)
    if n_rot:
        c = pairs(n_rot, 1)
        dw = noise * np.stack([normal(seed, S_T, 1000 + np.arange(n_rot), k) for k in range(3)], axis=1)
        Rrel = so3_exp(dw) @ (R[c[:, 0]].transpose(0, 2, 1) @ R[c[:, 1]])           # (T_a^-1 T_b).rotation
        T.rot_cams, T.rot_q, T.rot_w = c, quat_from_R(Rrel).astype(np.float32), np.full(n_rot, weight, np.float32)
    if n_xf:
        c = pairs(n_xf, 2)
        # C = T_b * T_a^-1 makes the error log(T_b^-1 C T_a) vanish on the ground truth
        Rc = R[c[:, 1]] @ R[c[:, 0]].transpose(0, 2, 1)
        pc = t[c[:, 1]] - np.einsum("nij,nj->ni", Rc, t[c[:, 0]])
        dw = noise * np.stack([normal(seed, S_T, 2000 + np.arange(n_xf), k) for k in range(3)], axis=1)
        dp = noise * np.stack([normal(seed, S_T, 2000 + np.arange(n_xf), 3 + k) for k in range(3)], axis=1)
        T.xf_cams, T.xf_p, T.xf_q = c, (pc + dp).astype(np.float32), quat_from_R(so3_exp(dw) @ Rc).astype(np.float32)
        T.xf_w = np.full(n_xf, weight * weight, np.float32)
    return T


def feed_tethers(bundler, T: "Tethers") -> None:
    """The BundlerLib call sequence of BundleAdjust.cpp:155-192 for a Tethers record (None = monocular map: three empty allocations)."""
    T = T if T is not None else Tethers()
    bundler.AllocateFixedDistanceConstraints(len(T.dist_d))
    for i in range(len(T.dist_d)):
        bundler.SetFixedDistanceConstraint(i, int(T.dist_cams[i, 0]), int(T.dist_cams[i, 1]), float(T.dist_d[i]), float(T.dist_w[i]))
    bundler.AllocateRelativeRotationConstraints(len(T.rot_w))
    for i in range(len(T.rot_w)):
        bundler.SetRelativeRotationConstraint(i, int(T.rot_cams[i, 0]), int(T.rot_cams[i, 1]), T.rot_q[i], float(T.rot_w[i]))
    bundler.AllocateRelativeTransformConstraints(len(T.xf_w))
    for i in range(len(T.xf_w)):
        bundler.SetRelativeTransformConstraint(i, int(T.xf_cams[i, 0]), int(T.xf_cams[i, 1]), T.xf_p[i], T.xf_q[i], float(T.xf_w[i]))


WIDTH, HEIGHT, FOCAL, CX, CY = 640, 480, 500.0, 320.0, 240.0
MARGIN = 8.0


def make_scene(n_cams: int, n_pts: int, n_obs: int, *, seed: int = 0x5EED0000, spacing: float | None = None,
               outlier_frac: float = 0.0, fixed: tuple[int, ...] = (0, 1), noise_px: float = 1.0,
               cam_sigma: float = 0.05, rot_sigma: float = 0.01, pt_sigma: float = 0.05) -> Scene:
    """Build the SURVEY 8d scene: a camera rail along +x looking down +z, points anchored per camera."""
    assert n_obs % n_pts == 0, "n_obs must be a multiple of n_pts (K observations per point)"
    K = n_obs // n_pts
    if spacing is None:
        spacing = 0.25 if n_cams > 10 else 0.05
    cam_idx = np.arange(n_cams)
    yaw = 0.05 * normal(seed, _S_YAW, cam_idx, 0)
    Rcw = _rot_y(yaw)                               # camera->world
    C = np.stack([cam_idx * spacing, np.zeros(n_cams), np.zeros(n_cams)], axis=1)
    R = Rcw.transpose(0, 2, 1)                      # world->camera
    t = -np.einsum("nij,nj->ni", R, C)

    # scan order of cameras around the anchor: a, a+1, a-1, a+2, a-2, ...
    max_off = 4 * K
    offs = [0]
    for d in range(1, max_off + 1):
        offs += [d, -d]
    offs = np.array(offs)

    pts_w = np.zeros((n_pts, 3))
    obs_cam = np.zeros((n_pts, K), dtype=np.int64)
    obs_uv = np.zeros((n_pts, K, 2))
    def _place(todo: np.ndarray, attempt: int) -> np.ndarray:
        """Try to place the points `todo` (attempt-th draw); returns the mask of those placed."""
        a = (todo * n_cams) // n_pts
        base = np.uint64(attempt * 3)
        u = MARGIN + (WIDTH - 2 * MARGIN) * uniform(seed, _S_PT, todo, base)
        v = MARGIN + (HEIGHT - 2 * MARGIN) * uniform(seed, _S_PT, todo, base + np.uint64(1))
        z = 3.0 + 7.0 * uniform(seed, _S_PT, todo, base + np.uint64(2))
        Xc = np.stack([(u - CX) / FOCAL * z, (v - CY) / FOCAL * z, z], axis=1)
        Xw = np.einsum("nij,nj->ni", Rcw[a], Xc) + C[a]
        cams = a[:, None] + offs[None, :]                       # (m, n_off)
        valid = (cams >= 0) & (cams < n_cams)
        cc = np.clip(cams, 0, n_cams - 1)
        Xcc = np.einsum("mkij,mj->mki", R[cc], Xw) + t[cc]
        zz = Xcc[..., 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            pu = FOCAL * Xcc[..., 0] / zz + CX
            pv = FOCAL * Xcc[..., 1] / zz + CY
        vis = valid & (zz > 0.5) & (pu >= MARGIN) & (pu <= WIDTH - MARGIN) & (pv >= MARGIN) & (pv <= HEIGHT - MARGIN)
        rank = np.cumsum(vis, axis=1)
        ok = rank[:, -1] >= K
        sel = vis & (rank <= K)
        good = np.nonzero(ok)[0]
        if good.size:
            _, cols = np.nonzero(sel[good])                     # first K visible in scan order
            cols = cols.reshape(good.size, K)
            g = todo[good]
            pts_w[g] = Xw[good]
            obs_cam[g] = np.take_along_axis(cc[good], cols, axis=1)
            obs_uv[g, :, 0] = np.take_along_axis(pu[good], cols, axis=1)
            obs_uv[g, :, 1] = np.take_along_axis(pv[good], cols, axis=1)
        return ok

    todo = np.arange(n_pts)
    attempt = 0
    while todo.size:
        assert attempt < 64, "scene generator failed to place points"
        failed = []
        for chunk in np.array_split(todo, max(1, -(-todo.size // 8192))):
            failed.append(chunk[~_place(chunk, attempt)])
        todo = np.concatenate(failed)
        attempt += 1

    obs_pt = np.repeat(np.arange(n_pts), K)
    obs_cam = obs_cam.reshape(-1)
    obs_uv = obs_uv.reshape(-1, 2)
    oidx = np.arange(n_obs)
    obs_uv = obs_uv + noise_px * np.stack([normal(seed, _S_NOISE, oidx, 0), normal(seed, _S_NOISE, oidx, 1)], axis=1)
    outlier_mask = np.zeros(n_obs, dtype=bool)
    if outlier_frac > 0:
        outlier_mask = uniform(seed, _S_OUTLIER, oidx, 0) < outlier_frac
        ou = WIDTH * uniform(seed, _S_OUTLIER, oidx, 1)
        ov = HEIGHT * uniform(seed, _S_OUTLIER, oidx, 2)
        obs_uv = np.where(outlier_mask[:, None], np.stack([ou, ov], axis=1), obs_uv)
    info = 1.0 - 1.0 / (1.5 + (obs_pt % 6)) ** 2       # Map/MappingMath.h:42-49 with n = p mod 6

    # perturbed initial state
    dC = cam_sigma * np.stack([normal(seed, _S_CAMPERT, cam_idx, k) for k in range(3)], axis=1)
    dw = rot_sigma * np.stack([normal(seed, _S_CAMPERT, cam_idx, 3 + k) for k in range(3)], axis=1)
    fixed_mask = np.zeros(n_cams, dtype=bool)
    fixed_mask[list(fixed)] = True
    dC[fixed_mask] = 0.0
    dw[fixed_mask] = 0.0
    R0 = so3_exp(dw) @ R
    t0 = -np.einsum("nij,nj->ni", R0, C + dC)
    pidx = np.arange(n_pts)
    P0 = pts_w + pt_sigma * np.stack([normal(seed, _S_PTPERT, pidx, k) for k in range(3)], axis=1)

    Kmat = np.tile(np.array([CX, CY, FOCAL, FOCAL], dtype=np.float32), (n_cams, 1))
    return Scene(
        n_cams=n_cams, n_pts=n_pts, n_obs=n_obs,
        cam_t=t0.astype(np.float32), cam_R=R0.astype(np.float32), cam_K=Kmat, cam_fixed=fixed_mask,
        points=P0.astype(np.float32), obs_uv=obs_uv.astype(np.float32),
        obs_cam=obs_cam.astype(np.uint32), obs_pt=obs_pt.astype(np.uint32), obs_info=info.astype(np.float32),
        gt_cam_t=t, gt_cam_R=R, gt_points=pts_w, outlier_mask=outlier_mask,
    )


# The five BASELINE.json configurations (config id -> generator arguments).
CONFIGS = {
    "tiny":   dict(n_cams=10, n_pts=200, n_obs=2000, seed=0x5EED0001),
    "local":  dict(n_cams=20, n_pts=5000, n_obs=50000, seed=0x5EED0003, fixed=(0, 1, 15, 16, 17, 18, 19)),
    "global": dict(n_cams=1000, n_pts=100000, n_obs=1000000, seed=0x5EED0004),
}


def make_config(name: str, **overrides) -> Scene:
    kw = dict(CONFIGS[name])
    kw.update(overrides)
    return make_scene(**kw)


SCENE_MAGIC = b"MAGESCN1"


def save_scene(scene: Scene, path: str) -> None:
    """Writes the float32 problem as one flat little-endian file for the C++ drivers (tools/scene_io.h reads it):
    magic, u32 n_cams n_pts n_obs 0, then cam_t[3n] cam_R_colmajor[9n] cam_K[4n] fixed[u32 n] points[3m] uv[2k] cam[u32 k]
    pt[u32 k] info[k].  No tethers."""
    with open(path, "wb") as f:
        f.write(SCENE_MAGIC)
        f.write(np.array([scene.n_cams, scene.n_pts, scene.n_obs, 0], "<u4").tobytes())
        for a, dt in ((scene.cam_t, "<f4"), (scene.cam_R_colmajor(), "<f4"), (scene.cam_K, "<f4"), (scene.cam_fixed, "<u4"),
                      (scene.points, "<f4"), (scene.obs_uv, "<f4"), (scene.obs_cam, "<u4"), (scene.obs_pt, "<u4"), (scene.obs_info, "<f4")):
            f.write(np.ascontiguousarray(a).astype(dt).tobytes())
