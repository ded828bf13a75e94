"""One large map sharded by keyframe window (BASELINE.json configs[4]: "8k-pose map sharded by keyframe
window, RCCL pose-block all-reduce over xGMI").

The dense reduced camera system of an 8k-pose map is 48 000^2 f64 = 18 GB and is never formed.  The map is cut into windows
of consecutive keyframes; every window is a BundlerLib problem with exactly the reference's local-BA semantics
(Map/ThreadSafeMap.cpp:868-971): its own keyframes are free, it carries every map point they observe, and every OTHER
keyframe observing one of those points enters as a FIXED camera (the halo).  One outer iteration = every window takes its LM
iterations with the halo frozen, then the windows exchange their poses -- the one real exchange step of the path:

    every rank writes the poses of the windows it owns into the POSE BLOCK, n_cams x 8 float64 (qx qy qz qw tx ty tz 0: the
    solver's own state rows), zeros elsewhere; all-reduce(SUM) over RCCL (512 KB at 8k poses); every window re-seeds the
    cameras it does not own from the result.  With the HIP back-end the block lives in HBM and nothing is staged through the
    host (mage_ba_export_poses_device / mage_ba_import_poses_device, include/mage_ba.h); the C++ form of this driver is
    include/mage_window.h (mageslam_amd/csrc/window_host.hip) -- both give the same block bit for bit (tests).

This is block-Jacobi on windows: not the monolithic solve, but it has the same fixed point (every copy of a shared point sees
all of that point's observations), and, because a window's step depends only on its own state and the exchanged block, the
result is bit-identical for any assignment of windows to ranks (tests/test_windowed*.py).  Windows are independent between
exchanges, so a rank that owns several simply steps them one after another (or on concurrent handles).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .scene import Scene


@dataclass
class Window:
    index: int
    own: np.ndarray          # global camera indices that are free here (the window's own keyframes)
    cams: np.ndarray         # global camera indices of the sub-problem: own, then overlap (free), then halo (fixed)
    pts: np.ndarray          # global point indices (observed by at least one free keyframe)
    obs: np.ndarray          # global observation indices (every observation of those points)
    scene: Scene             # the sub-problem in local indices


def cut_windows(scene: Scene, n_windows: int, overlap: int = 0) -> list[Window]:
    """Windows of consecutive keyframes with their halo, as local-index sub-scenes.  With `overlap` > 0 the `overlap`
    keyframes either side of a window are free in it too (restricted additive Schwarz: they move with the window, but only
    the owner's value of a keyframe is ever published), which lets corrections cross a window boundary within one outer
    iteration instead of one boundary per iteration."""
    nc = scene.n_cams
    bounds = [(w * nc) // n_windows for w in range(n_windows + 1)]
    obs_cam = scene.obs_cam.astype(np.int64); obs_pt = scene.obs_pt.astype(np.int64)
    out = []
    for w in range(n_windows):
        lo, hi = bounds[w], bounds[w + 1]
        flo, fhi = max(0, lo - overlap), min(nc, hi + overlap)                 # free keyframes: own + overlap
        free_mask_obs = (obs_cam >= flo) & (obs_cam < fhi)
        pt_sel = np.zeros(scene.n_pts, bool); pt_sel[obs_pt[free_mask_obs]] = True
        obs_idx = np.nonzero(pt_sel[obs_pt])[0]                                  # map order is kept
        cam_sel = np.zeros(nc, bool); cam_sel[obs_cam[obs_idx]] = True; cam_sel[flo:fhi] = True
        own = np.arange(lo, hi)
        ovl = np.concatenate([np.arange(flo, lo), np.arange(hi, fhi)])
        inside = (np.arange(nc) >= flo) & (np.arange(nc) < fhi)
        halo = np.nonzero(cam_sel & ~inside)[0]
        cams = np.concatenate([own, ovl, halo])
        pts = np.nonzero(pt_sel)[0]
        cam_l = np.full(nc, -1, np.int64); cam_l[cams] = np.arange(len(cams))
        pt_l = np.full(scene.n_pts, -1, np.int64); pt_l[pts] = np.arange(len(pts))
        fixed = np.concatenate([scene.cam_fixed[own], scene.cam_fixed[ovl], np.ones(len(halo), bool)])
        sub = Scene(n_cams=len(cams), n_pts=len(pts), n_obs=len(obs_idx), cam_t=scene.cam_t[cams], cam_R=scene.cam_R[cams],
                    cam_K=scene.cam_K[cams], cam_fixed=fixed, points=scene.points[pts], obs_uv=scene.obs_uv[obs_idx],
                    obs_cam=cam_l[obs_cam[obs_idx]].astype(np.uint32), obs_pt=pt_l[obs_pt[obs_idx]].astype(np.uint32),
                    obs_info=scene.obs_info[obs_idx])
        out.append(Window(w, own, cams, pts, obs_idx, sub))
    return out


def owned_windows(n_windows: int, rank: int, world: int) -> list[int]:
    """Contiguous blocks of windows per rank (neighbouring windows share most of their halo)."""
    return [w for w in range(n_windows) if (w * world) // n_windows == rank]


class WindowedMap:
    """Drives the windows owned by this rank.  `make_bundler()` returns a BundlerLib-surface object (the HIP back-end in
    production; tests also pass the CPU oracle to pin the driver's logic); `dist` is torch.distributed or None.

    With the HIP back-end (`device` = HIP ordinal) the pose block is a float64 torch tensor in HBM: windows export / import
    their rows with kernels ordered on torch's current stream, and the all-reduce runs on the device tensor (RCCL); only a
    gloo control plane stages the 8 * n_cams doubles through the host.  With any other back-end (the oracle) the block is a
    numpy array."""

    def __init__(self, scene: Scene, n_windows: int, make_bundler, load, *, rank: int = 0, world: int = 1, dist=None,
                 exchange_device: str = "cpu", overlap: int = 0, threads: int = 1, device: int | None = None):
        self.scene, self.dist, self.rank, self.world = scene, dist, rank, world
        self.exchange_device = exchange_device
        self.threads = threads
        self.windows = cut_windows(scene, n_windows, overlap)
        self.mine = owned_windows(n_windows, rank, world)
        self.bundlers = {}
        for w in self.mine:
            b = make_bundler()
            load(b, self.windows[w].scene)
            self.bundlers[w] = b
        self.on_device = bool(self.mine) and hasattr(self.bundlers[self.mine[0]], "ExportPosesDevice") and device is not None
        self.block = None                                 # host copy of the pose block as of the last exchange (None: fetch / none yet)
        self.exchanged = False                            # the block is the MAP only after the first exchange (before it: nothing published)
        self.removed = {w: 0 for w in self.mine}          # observations the outlier passes of earlier steps took out, per window
        if self.on_device:
            import torch
            self._torch = torch
            self.device = torch.device("cuda", device)
            self.tblock = torch.zeros((scene.n_cams, 8), dtype=torch.float64, device=self.device)
            for w in self.mine:
                win = self.windows[w]
                k = len(win.own)
                self.bundlers[w].BindPoseExchange(np.arange(k), win.own, np.arange(k, len(win.cams)), win.cams[k:])
        self.exchanged_bytes = 0

    def outer_iteration(self, huber: float, max_err_sq: float = 1e30, inner: int = 1) -> float:
        """Every owned window takes `inner` LM iterations against the frozen halo, then poses are exchanged and the halos
        re-seeded.  Returns the observation-weighted mean square error over the owned windows (before the exchange)."""
        def step(w):
            out: list = []
            mse = self.bundlers[w].StepBundleAdjustment([huber] * inner, max_err_sq, out)
            # the step's mean is over what was active when it ran minus what it classified as outliers (BundlerLib.cpp:386-446)
            n = self.windows[w].scene.n_obs - self.removed[w] - len(out)
            self.removed[w] += len(out)
            return float(mse), n

        # The windows of a rank are independent between exchanges: stepped from concurrent host threads, each handle on its
        # own stream, the chain-bound tail of one factorisation overlaps the matrix-core bulk of another.
        if self.threads > 1 and len(self.mine) > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(self.threads, len(self.mine))) as ex:
                results = list(ex.map(step, self.mine))
        else:
            results = [step(w) for w in self.mine]
        err_sum, n_sum = 0.0, 0
        for mse, n in results:
            if n > 0 and np.isfinite(mse):
                err_sum += mse * n; n_sum += n
        self.exchange()
        return err_sum / n_sum if n_sum else float("nan")

    def exchange(self) -> None:
        self.exchanged = True
        if self.on_device:
            return self._exchange_device()
        block = np.zeros((self.scene.n_cams, 8))
        for w in self.mine:
            win = self.windows[w]
            k = len(win.own)
            block[win.own, :7] = self.bundlers[w].poses_f64()[:k]
        block += 0.0                          # -0.0 -> +0.0, which is what a sum with the other ranks' zeros does anyway
        if self.dist is not None:
            import torch
            buf = torch.from_numpy(block).to(self.exchange_device)
            self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM)          # disjoint rows + zeros: the sum is exact
            block = buf.cpu().numpy()
            self.exchanged_bytes += block.nbytes
        self.block = block
        for w in self.mine:
            win = self.windows[w]
            k = len(win.own)
            rest = win.cams[k:]                   # everything this window does not own: overlap keyframes and the fixed halo
            if len(rest):
                b = self.bundlers[w]
                lam = b.GetCurrentLambda()
                b.SetCameraPosesF64(np.arange(k, len(win.cams), dtype=np.uint32), block[rest])
                if lam > 0:
                    b.SetCurrentLambda(lam)       # the damping carries over, as MappingWorker carries it from one BA to the next

    def _exchange_device(self) -> None:
        torch = self._torch
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream().cuda_stream
            self.tblock.zero_()
            ptr = self.tblock.data_ptr()
            for w in self.mine:
                self.bundlers[w].ExportPosesDevice(ptr, st)
            if self.dist is not None:
                if self.dist.get_backend() == "nccl":
                    self.dist.all_reduce(self.tblock, op=self.dist.ReduceOp.SUM)      # RCCL on the device block
                else:                                                                  # gloo control plane: 8 n doubles through the host
                    buf = self.tblock.cpu()
                    self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM)
                    self.tblock.copy_(buf)
                self.exchanged_bytes += self.tblock.numel() * 8
            for w in self.mine:
                b = self.bundlers[w]
                lam = b.GetCurrentLambda()
                b.ImportPosesDevice(ptr, st)
                if lam > 0:
                    b.SetCurrentLambda(lam)
            self.block = None                     # fetched on demand

    def pose_block(self) -> np.ndarray:
        """(n_cams, 8) float64 rows qx qy qz qw tx ty tz 0 of the whole map as of the last exchange."""
        if not self.exchanged:
            # before the first exchange nothing has been published: zero rows would read as identity rotations at the origin
            raise RuntimeError("the pose block is defined after the first exchange (outer_iteration() or exchange())")
        if self.block is None:
            self.block = self.tblock.cpu().numpy()
        return self.block

    def poses(self):
        """(t, R column-major) float32 of the whole map as of the last exchange (the BundlerLib surface's pose format)."""
        b = self.pose_block()
        x, y, z, w = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
        R = np.empty((len(b), 3, 3))
        R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - z * w); R[:, 0, 2] = 2 * (x * z + y * w)
        R[:, 1, 0] = 2 * (x * y + z * w); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - x * w)
        R[:, 2, 0] = 2 * (x * z - y * w); R[:, 2, 1] = 2 * (y * z + x * w); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
        return b[:, 4:7].astype(np.float32), np.ascontiguousarray(R.transpose(0, 2, 1).reshape(len(b), 9)).astype(np.float32)
