"""Independent numpy/scipy bundle adjuster -- TEST INFRASTRUCTURE ONLY (never shipped, never timed).

Purpose: pin oracle/ba_oracle.c ("independent double implementation").  It shares no
code with the C oracle and takes different routes everywhere it can:

  * rotations are 3x3 matrices (not quaternions); the update is R <- Exp(w) R with Rodrigues;
  * Jacobians come from the chain rule  J = -dpi/dXc * dXc/d(.)  with dXc/dw = -[Xc]x, dXc/dv = I,
    dXc/dX = R (g2o hand-expands these entries; appendix A.2);
  * the damped normal equations are solved as ONE sparse system over poses and points with
    scipy.sparse.linalg.spsolve (no Schur complement, no LDLT);
  * everything is vectorised over edges.

The Levenberg-Marquardt control flow (lambda init tau*max diag, rho test, 1e-3, 2/3, 1/3, ni doubling,
<=10 trials) and the StepBundleAdjustment post-pass follow SURVEY.md appendix A.4 and
Dependencies/BundlerLib/Source/BundlerLib.cpp:364-447, because that is the behaviour being pinned.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def _skew(v):
    K = np.zeros(v.shape[:-1] + (3, 3))
    K[..., 0, 1] = -v[..., 2]; K[..., 0, 2] = v[..., 1]
    K[..., 1, 0] = v[..., 2]; K[..., 1, 2] = -v[..., 0]
    K[..., 2, 0] = -v[..., 1]; K[..., 2, 1] = v[..., 0]
    return K


def _se3_exp(u):
    """u = [omega | upsilon] -> (R, t); closed forms, vectorised."""
    w, v = u[..., :3], u[..., 3:]
    th = np.linalg.norm(w, axis=-1)
    small = th < 1e-5
    ths = np.where(small, 1.0, th)
    K = _skew(w)
    K2 = K @ K
    A = np.where(small, 1.0, np.sin(ths) / ths)[..., None, None]
    B = np.where(small, 0.5, (1 - np.cos(ths)) / ths ** 2)[..., None, None]
    Cc = np.where(small, 1.0 / 6.0, (ths - np.sin(ths)) / ths ** 3)[..., None, None]
    I = np.eye(3)
    R = I + A * K + B * K2
    V = I + B * K + Cc * K2
    return R, np.einsum("...ij,...j->...i", V, v)


def _quatf_from_R(Rf):
    """float32 matrix->quaternion->normalise, then back to a float64 rotation (BundlerLib.cpp:272)."""
    Rf = Rf.astype(np.float32)
    n = Rf.shape[0]
    q = np.zeros((n, 4), np.float32)  # x y z w
    for i in range(n):
        m = Rf[i]
        tr = np.float32(m[0, 0] + m[1, 1] + m[2, 2])
        if tr > 0:
            s = np.sqrt(np.float32(tr + np.float32(1)))
            q[i, 3] = np.float32(0.5) * s
            s = np.float32(0.5) / s
            q[i, 0] = (m[2, 1] - m[1, 2]) * s; q[i, 1] = (m[0, 2] - m[2, 0]) * s; q[i, 2] = (m[1, 0] - m[0, 1]) * s
        else:
            a = 0
            if m[1, 1] > m[0, 0]: a = 1
            if m[2, 2] > m[a, a]: a = 2
            b = (a + 1) % 3; c = (b + 1) % 3
            s = np.sqrt(np.float32(m[a, a] - m[b, b] - m[c, c] + np.float32(1)))
            q[i, a] = np.float32(0.5) * s
            s = np.float32(0.5) / s
            q[i, 3] = (m[c, b] - m[b, c]) * s; q[i, b] = (m[b, a] + m[a, b]) * s; q[i, c] = (m[c, a] + m[a, c]) * s
        q[i] /= np.sqrt(np.float32(np.sum(q[i] * q[i], dtype=np.float32)))
    q = q.astype(np.float64)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((n, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - z * w); R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w); R[:, 2, 1] = 2 * (y * z + x * w); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _rot_from_quat(q):
    """x, y, z, w (any norm) -> rotation matrix."""
    q = np.asarray(q, np.float64)
    q = q / np.linalg.norm(q)
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _rot_angle(M):
    """Rotation angle in [0, pi], atan2 form (well conditioned near 0, unlike acos of the trace)."""
    v = np.array([M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1]])
    return np.arctan2(0.5 * np.linalg.norm(v), 0.5 * (np.trace(M) - 1.0))


def _se3_log(R, t):
    """[omega | upsilon] of the rigid transform (R, t): axis-angle, then V^-1 t."""
    v = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])   # sin(theta) * axis
    c = 0.5 * (np.trace(R) - 1.0)
    if abs(c) > 0.99999:
        w = v
        K = _skew(w)
        Vi = np.eye(3) - 0.5 * K + K @ K / 12.0
    else:
        th = np.arccos(c)
        w = v * th / np.sqrt(1 - c * c)
        K = _skew(w)
        Vi = np.eye(3) - 0.5 * K + (1 - th / (2 * np.tan(th / 2))) / th ** 2 * (K @ K)
    return np.concatenate([w, Vi @ t])


def _adj(R, t):
    A = np.zeros((6, 6))
    A[:3, :3] = R; A[3:, 3:] = R; A[3:, :3] = _skew(t) @ R
    return A


class _Tether:
    """One pose-pose constraint; residual as a function of the two poses, 4x4-matrix algebra throughout."""
    def __init__(self, kind, a, b, w, dist=0.0, q=None, p=None):
        self.kind, self.a, self.b, self.w, self.dist = kind, int(a), int(b), float(w), float(dist)
        if kind == "rot":
            self.Rm = _rot_from_quat(q)
        if kind == "xf":
            self.C = np.eye(4); self.C[:3, :3] = _rot_from_quat(q); self.C[:3, 3] = np.asarray(p, np.float64)
        self.dim = 6 if kind == "xf" else 1
        self.omega = self.w if kind == "xf" else 1.0       # information: w * I6 for the transform edge, 1 otherwise

    def residual(self, Ra, ta, Rb, tb):
        if self.kind == "dist":
            return np.array([(self.dist - np.linalg.norm(tb - ta)) * self.w])
        if self.kind == "rot":
            return np.array([_rot_angle(Ra.T @ Rb @ self.Rm.T) * self.w])
        Ta = np.eye(4); Ta[:3, :3] = Ra; Ta[:3, 3] = ta
        Tb = np.eye(4); Tb[:3, :3] = Rb; Tb[:3, 3] = tb
        E = np.linalg.inv(Tb) @ self.C @ Ta
        return _se3_log(E[:3, :3], E[:3, 3])

    def jacobians(self, Ra, ta, Rb, tb, fixed_a, fixed_b):
        Ja, Jb = np.zeros((self.dim, 6)), np.zeros((self.dim, 6))
        if self.kind == "xf":       # g2o's closed form: Adj(Tb^-1 C), -Adj(Ta^-1 C^-1)
            Ta = np.eye(4); Ta[:3, :3] = Ra; Ta[:3, 3] = ta
            Tb = np.eye(4); Tb[:3, :3] = Rb; Tb[:3, 3] = tb
            A = np.linalg.inv(Tb) @ self.C
            B = np.linalg.inv(Ta) @ np.linalg.inv(self.C)
            return _adj(A[:3, :3], A[:3, 3]), -_adj(B[:3, :3], B[:3, 3])
        h = 1e-9                    # g2o BaseMultiEdge: central differences through the manifold update, fixed vertices skipped
        for k in range(6):
            u = np.zeros(6); u[k] = h
            dRp, dtp = _se3_exp(u); dRm, dtm = _se3_exp(-u)
            if not fixed_a:
                Ja[:, k] = (self.residual(dRp @ Ra, dRp @ ta + dtp, Rb, tb) - self.residual(dRm @ Ra, dRm @ ta + dtm, Rb, tb)) / (2 * h)
            if not fixed_b:
                Jb[:, k] = (self.residual(Ra, ta, dRp @ Rb, dRp @ tb + dtp) - self.residual(Ra, ta, dRm @ Rb, dRm @ tb + dtm)) / (2 * h)
        return Ja, Jb


class NumpyBundler:
    def __init__(self, scene, points_fixed=False):
        self.tethers = []
        T = getattr(scene, "tethers", None)
        if T is not None:
            for i in range(len(T.dist_d)):
                self.tethers.append(_Tether("dist", T.dist_cams[i, 0], T.dist_cams[i, 1], np.float32(T.dist_w[i]), dist=np.float32(T.dist_d[i])))
            for i in range(len(T.rot_w)):
                self.tethers.append(_Tether("rot", T.rot_cams[i, 0], T.rot_cams[i, 1], np.float32(T.rot_w[i]), q=T.rot_q[i].astype(np.float32)))
            for i in range(len(T.xf_w)):
                self.tethers.append(_Tether("xf", T.xf_cams[i, 0], T.xf_cams[i, 1], np.float32(T.xf_w[i]), q=T.xf_q[i].astype(np.float32), p=T.xf_p[i].astype(np.float32)))
        self.R = _quatf_from_R(scene.cam_R)                       # (nc,3,3) world->camera
        self.t = scene.cam_t.astype(np.float64)
        self.f = scene.cam_K[:, 2].astype(np.float64)
        self.c = scene.cam_K[:, :2].astype(np.float64)
        self.cam_fixed = scene.cam_fixed.copy()
        self.X = scene.points.astype(np.float64)
        self.uv = scene.obs_uv.astype(np.float64)
        self.cam = scene.obs_cam.astype(np.int64)
        self.pt = scene.obs_pt.astype(np.int64)
        self.info = scene.obs_info.astype(np.float64)
        self.points_fixed = points_fixed
        self.removed = np.zeros(self.uv.shape[0], bool)
        self.dirty = True
        self.iteration = 0
        self.lam = -1.0
        self.user_lambda = 0.0
        self.ni = 2.0
        self.err = np.zeros_like(self.uv)
        self.trace = []

    def SetCurrentLambda(self, l):
        self.iteration = 0
        self.user_lambda = float(np.float32(l))

    # -- structure
    def _init(self):
        act = ~self.removed
        if self.points_fixed:
            act &= ~self.cam_fixed[self.cam]
        self.act = np.nonzero(act)[0]
        nc, npnt = self.R.shape[0], self.X.shape[0]
        cam_deg = np.bincount(self.cam[self.act], minlength=nc)
        self.act_teth = [T for T in self.tethers if not (self.cam_fixed[T.a] and self.cam_fixed[T.b])]
        for T in self.act_teth:
            cam_deg[T.a] += 1; cam_deg[T.b] += 1
        pt_deg = np.bincount(self.pt[self.act], minlength=npnt)
        free_c = (~self.cam_fixed) & (cam_deg > 0)
        free_p = (pt_deg > 0) & (not self.points_fixed)
        self.hc = np.full(nc, -1); self.hc[free_c] = np.arange(free_c.sum())
        self.hp = np.full(npnt, -1); self.hp[free_p] = np.arange(free_p.sum())
        self.nfc, self.nfp = int(free_c.sum()), int(free_p.sum())
        self.free_c, self.free_p = np.nonzero(free_c)[0], np.nonzero(free_p)[0]
        self.useless = (self.nfc + self.nfp) == 0
        self.iteration = 0
        self.dirty = False

    def _residuals(self, R, t, X):
        e = self.act
        Xc = np.einsum("nij,nj->ni", R[self.cam[e]], X[self.pt[e]]) + t[self.cam[e]]
        proj = Xc[:, :2] / Xc[:, 2:3] * self.f[self.cam[e], None] + self.c[self.cam[e]]
        return self.uv[e] - proj, Xc

    def _chi(self, r, delta):
        chi2 = self.info[self.act] * np.sum(r * r, axis=1)
        d2 = delta * delta
        s = np.sqrt(np.maximum(chi2, 1e-300))
        rho0 = np.where(chi2 <= d2, chi2, 2 * s * delta - d2)
        rho1 = np.where(chi2 <= d2, 1.0, delta / s)
        return rho0.sum(), rho1

    def _teth_chi(self, R, t):
        return sum(T.omega * float(np.sum(T.residual(R[T.a], t[T.a], R[T.b], t[T.b]) ** 2)) for T in self.act_teth)

    def _lm(self, delta):
        e = self.act
        r, Xc = self._residuals(self.R, self.t, self.X)
        self.err[e] = r
        cur, rho1 = self._chi(r, delta)
        cur += self._teth_chi(self.R, self.t)
        chi_before = cur
        f = self.f[self.cam[e]]
        x, y, z = Xc[:, 0], Xc[:, 1], Xc[:, 2]
        dpi = np.zeros((e.size, 2, 3))
        dpi[:, 0, 0] = f / z; dpi[:, 0, 2] = -f * x / z ** 2
        dpi[:, 1, 1] = f / z; dpi[:, 1, 2] = -f * y / z ** 2
        dXc_dpose = np.concatenate([-_skew(Xc), np.broadcast_to(np.eye(3), (e.size, 3, 3))], axis=2)  # (m,3,6)
        Jc = -dpi @ dXc_dpose                         # (m,2,6)
        Jp = -dpi @ self.R[self.cam[e]]               # (m,2,3)
        w = self.info[e] * rho1
        hc, hp = self.hc[self.cam[e]], self.hp[self.pt[e]]
        n = 6 * self.nfc + 3 * self.nfp
        # sparse Jacobian of sqrt-weighted residuals
        rows, cols, vals = [], [], []
        ridx = np.arange(e.size)
        sw = np.sqrt(w)
        mc = hc >= 0
        for a in range(2):
            for k in range(6):
                rows.append(2 * ridx[mc] + a); cols.append(6 * hc[mc] + k); vals.append(sw[mc] * Jc[mc, a, k])
        mp = hp >= 0
        for a in range(2):
            for k in range(3):
                rows.append(2 * ridx[mp] + a); cols.append(6 * self.nfc + 3 * hp[mp] + k); vals.append(sw[mp] * Jp[mp, a, k])
        J = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(2 * e.size, n))
        H = (J.T @ J).tocsc()
        g = -(J.T @ (sw[:, None] * r).reshape(-1))     # b = -J^T W r
        if self.act_teth:                               # dense rows of the pose-pose constraints
            H = H.tolil()
            for T in self.act_teth:
                res = T.residual(self.R[T.a], self.t[T.a], self.R[T.b], self.t[T.b])
                Ja, Jb = T.jacobians(self.R[T.a], self.t[T.a], self.R[T.b], self.t[T.b], self.cam_fixed[T.a], self.cam_fixed[T.b])
                ha, hb = self.hc[T.a], self.hc[T.b]
                for (hi, Ji) in ((ha, Ja), (hb, Jb)):
                    if hi >= 0:
                        H[6 * hi:6 * hi + 6, 6 * hi:6 * hi + 6] += T.omega * (Ji.T @ Ji)
                        g[6 * hi:6 * hi + 6] -= T.omega * (Ji.T @ res)
                if ha >= 0 and hb >= 0:
                    H[6 * ha:6 * ha + 6, 6 * hb:6 * hb + 6] += T.omega * (Ja.T @ Jb)
                    H[6 * hb:6 * hb + 6, 6 * ha:6 * ha + 6] += T.omega * (Jb.T @ Ja)
            H = H.tocsc()
        if self.iteration == 0:
            self.lam = self.user_lambda if self.user_lambda > 0 else 1e-5 * np.abs(H.diagonal()).max()
            self.ni = 2.0
        rho = 0.0
        q = 0
        while True:
            Hd = H + self.lam * sp.identity(n, format="csc")
            try:
                dx = spla.spsolve(Hd, g)
                ok = np.all(np.isfinite(dx))
            except Exception:
                ok = False; dx = np.zeros(n)
            Rn, tn, Xn = self.R.copy(), self.t.copy(), self.X.copy()
            if self.nfc:
                dR, dt = _se3_exp(dx[: 6 * self.nfc].reshape(-1, 6))
                fc = self.free_c
                Rn[fc] = dR @ self.R[fc]
                tn[fc] = np.einsum("nij,nj->ni", dR, self.t[fc]) + dt
            if self.nfp:
                Xn[self.free_p] += dx[6 * self.nfc:].reshape(-1, 3)
            rn, _ = self._residuals(Rn, tn, Xn)
            self.err[e] = rn
            tmp, _ = self._chi(rn, delta)
            tmp += self._teth_chi(Rn, tn)
            if not ok:
                tmp = np.finfo(np.float64).max
            scale = float(np.sum(dx * (self.lam * dx + g))) + 1e-3
            rho = (cur - tmp) / scale
            if rho > 0 and np.isfinite(tmp):
                alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                self.lam *= max(1.0 / 3.0, alpha)
                self.ni = 2.0
                cur = tmp
                self.R, self.t, self.X = Rn, tn, Xn
            else:
                self.lam *= self.ni
                self.ni *= 2
            q += 1
            if not (rho < 0 and q < 10):
                break
        code = 1 if (q == 10 or rho == 0) else 0
        self.trace.append(dict(code=code, trials=q, chi_before=chi_before, chi_after=cur, lam=self.lam))
        return code

    def _step(self, delta):
        if self.dirty:
            self._init()
        if self.useless:
            return False
        code = self._lm(delta)
        self.iteration += 1
        return code == 0

    def StepBundleAdjustment(self, huber_widths, max_err_sq, outliers):
        self.trace = []
        for hw in huber_widths:
            if not self._step(float(np.float32(hw))):
                break
        e = self.act
        ss = np.sum(self.err[e] ** 2, axis=1)
        # in front of the camera: (X - C) . (R^T ez) = (R X + t)_z
        zc = (np.einsum("nij,nj->ni", self.R[self.cam[e]], self.X[self.pt[e]]) + self.t[self.cam[e]])[:, 2]
        bad = (zc <= 0) | (ss > float(np.float32(max_err_sq)))
        out = e[bad]
        self.removed[out] = True
        if out.size:
            self.dirty = True
        outliers.extend(int(i) for i in out)
        good = ~bad
        with np.errstate(invalid="ignore", divide="ignore"):
            return float(np.float32(ss[good].sum() / good.sum()))
