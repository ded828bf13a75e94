"""Landmark-sharded solve of ONE map on several GPUs -- the "exact algorithm" of SURVEY.md section 8e.

The reference has nothing like it (g2o runs one thread behind ``StepOptimizer::Step``, BundlerLib.cpp:132-149); the numbers it
has to reproduce are the single-handle solve's.  The split: every rank holds ALL cameras and its own share of the map points
with all their observations; per Levenberg-Marquardt trial the ranks add up their reduced camera systems (one all-reduce of the
packed lower tiles of S and the right-hand side, 148 MB at 1 000 poses), every rank factorises the sum, moves the cameras and
back-substitutes its own landmarks.  All of that happens inside ``mage_ba_step`` (include/mage_ba.h:
``mage_ba_set_landmark_shard``); this module is the host side around it:

* ``partition_landmarks`` -- who owns which map point, balanced by the Schur work a point causes (k (k + 1) / 2 blocks for k
  observations);
* ``shard_scene`` -- a rank's sub-problem in local indices (all cameras, its points, their observations, its share of the tethers);
* three all-reduce callbacks for the C ABI: ``ThreadGroup`` (one process drives all shards from threads -- several handles on one
  GPU; what the single-GPU tests use), ``TorchGroup`` (one process per GPU, ``torch.distributed``: RCCL on device memory, or gloo
  through the host as a control-plane fallback), and the C++ twin with ``ncclAllReduce`` in tools/sharded_rccl.cpp;
* ``ShardedBundler`` -- the BundlerLib surface of one rank with outliers and points reported in the MAP's indices.
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from ._lib import check, lib
from .scene import Scene, Tethers

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
OP_SUM, OP_MAX = 0, 1


def landmark_weights(obs_pt: np.ndarray, n_pts: int) -> np.ndarray:
    """Blocks of S a point contributes to: k (k + 1) / 2 for k observations ("balanced by sum k(k+1)/2")."""
    k = np.bincount(np.asarray(obs_pt, np.int64), minlength=n_pts).astype(np.int64)
    return k * (k + 1) // 2


def partition_landmarks(obs_pt: np.ndarray, n_pts: int, n_ranks: int) -> np.ndarray:
    """owner[n_pts] in [0, n_ranks): heaviest point first onto the lightest rank (ties: lower point index, lower rank), so the
    result depends on nothing but the arguments -- every rank computes the same table without talking to the others.  The table
    is the library's (``mage_ba_partition_landmarks``, host code): the C++ driver and this one cannot disagree."""
    if n_ranks < 1:
        raise ValueError("n_ranks must be positive")
    L = lib()
    L.mage_ba_partition_landmarks.argtypes = [C.c_size_t, C.c_size_t, np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS"), C.c_int,
                                              np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")]
    owner = np.zeros(max(n_pts, 1), np.int32)
    obs = np.ascontiguousarray(obs_pt, np.uint32)
    check(L.mage_ba_partition_landmarks(n_pts, len(obs), obs if len(obs) else np.zeros(1, np.uint32), n_ranks, owner))
    return owner[:n_pts]


def shard_scene(scene: Scene, owner: np.ndarray, rank: int, n_ranks: int):
    """(sub-scene of `rank`, global point index per local point, global observation index per local observation).  Cameras are
    the map's, unchanged and in order; tether edge k of each kind goes to rank k % n_ranks."""
    pts = np.nonzero(owner == rank)[0]
    obs = np.nonzero(owner[scene.obs_pt.astype(np.int64)] == rank)[0]
    pt_l = np.full(scene.n_pts, -1, np.int64); pt_l[pts] = np.arange(len(pts))
    teth = None
    if scene.tethers is not None:
        T = scene.tethers
        pick = lambda n: np.arange(n)[np.arange(n) % n_ranks == rank]      # noqa: E731
        d, r, x = pick(len(T.dist_d)), pick(len(T.rot_w)), pick(len(T.xf_w))
        teth = Tethers(dist_cams=T.dist_cams[d], dist_d=T.dist_d[d], dist_w=T.dist_w[d], rot_cams=T.rot_cams[r], rot_q=T.rot_q[r],
                       rot_w=T.rot_w[r], xf_cams=T.xf_cams[x], xf_p=T.xf_p[x], xf_q=T.xf_q[x], xf_w=T.xf_w[x])
    sub = Scene(n_cams=scene.n_cams, n_pts=len(pts), n_obs=len(obs), cam_t=scene.cam_t, cam_R=scene.cam_R, cam_K=scene.cam_K,
                cam_fixed=scene.cam_fixed, points=scene.points[pts], obs_uv=scene.obs_uv[obs], obs_cam=scene.obs_cam[obs],
                obs_pt=pt_l[scene.obs_pt[obs].astype(np.int64)].astype(np.uint32), obs_info=scene.obs_info[obs], tethers=teth)
    return sub, pts, obs


# ---------------------------------------------------------------------------------------------------------------------------
# all-reduce callbacks
# ---------------------------------------------------------------------------------------------------------------------------
_hip = None


def _hip_runtime():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    return _hip


class ThreadGroup:
    """All shards in ONE process, one host thread per rank (the ranks' steps must run concurrently: every exchange is a
    rendezvous).  A rank's callback drains its stream, the last to arrive adds the buffers with ``mage_device_allreduce_local``
    (sum in rank order) and everybody leaves when that kernel is done."""

    def __init__(self, n_ranks: int):
        self.n = n_ranks
        self._bar = threading.Barrier(n_ranks)
        self._bufs = [0] * n_ranks
        self._counts = [0] * n_ranks
        self._failed = False
        self.calls = 0
        self.doubles = 0
        L = lib()
        L.mage_device_allreduce_local.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_size_t, C.c_int, C.c_void_p]

    def callback(self, rank: int):
        hip, L = _hip_runtime(), lib()

        def fn(_user, buf, count, op, stream):
            try:
                if hip.hipStreamSynchronize(stream) != 0:
                    self._failed = True
                self._bufs[rank], self._counts[rank] = buf, count
                if self._bar.wait() == 0:
                    if len(set(self._counts)) != 1:
                        self._failed = True                 # the ranks disagree about the shape of the system
                    else:
                        arr = (C.c_void_p * self.n)(*self._bufs)
                        if L.mage_device_allreduce_local(arr, self.n, count, op, stream) != 0 or hip.hipStreamSynchronize(stream) != 0:
                            self._failed = True
                    self.calls += 1
                    self.doubles += count
                self._bar.wait()
                return 1 if self._failed else 0
            except threading.BrokenBarrierError:
                return 1
        return ALLREDUCE_FN(fn)

    def abort(self):
        self._bar.abort()


class _DeviceArray:
    """Just enough of the CUDA array interface for ``torch.as_tensor`` to alias `count` doubles at a device address."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


class TorchGroup:
    """One process per rank over ``torch.distributed``.  backend "nccl" (= RCCL): the all-reduce runs on the device buffer, in
    stream order with the handle's stream, no host synchronisation.  Any other backend (gloo: CPU tests, or a node whose RCCL
    does not come up): the buffer is staged through the host."""

    def __init__(self, dist, device_index: int | None):
        import torch
        self._torch, self.dist, self.device_index = torch, dist, device_index
        self.calls = 0
        self.doubles = 0

    def callback(self):
        torch, dist = self._torch, self.dist
        ops = {OP_SUM: dist.ReduceOp.SUM, OP_MAX: dist.ReduceOp.MAX}

        def fn(_user, buf, count, op, stream):
            try:
                self.calls += 1
                self.doubles += count
                if self.device_index is None:                       # host memory (CPU tests of the plumbing)
                    a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double)), shape=(count,))
                    dist.all_reduce(torch.from_numpy(a), op=ops[op])
                    return 0
                dev = torch.device("cuda", self.device_index)
                ext = torch.cuda.ExternalStream(stream, device=dev)
                with torch.cuda.stream(ext):
                    t = torch.as_tensor(_DeviceArray(buf, count), device=dev)
                    if dist.get_backend() == "nccl":
                        dist.all_reduce(t, op=ops[op])                # ordered with `stream` on both sides
                    else:
                        h = t.cpu()
                        dist.all_reduce(h, op=ops[op])
                        t.copy_(h)
                        ext.synchronize()
                return 0
            except Exception as e:  # noqa: BLE001 - a callback must not raise through C
                import sys
                print(f"[mageslam_amd.sharded] all-reduce failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
                return 1
        return ALLREDUCE_FN(fn)


# ---------------------------------------------------------------------------------------------------------------------------
# one rank's bundler
# ---------------------------------------------------------------------------------------------------------------------------
class ShardedBundler:
    """The BundlerLib surface of one rank of a landmark-sharded map.  `make_bundler()` returns the HIP back-end's BundlerLib;
    `callback` is an ALLREDUCE_FN (kept alive here).  Outliers come back as observation indices OF THE MAP (this rank's only);
    ``points_into`` scatters this rank's points into a map-sized array."""

    def __init__(self, scene: Scene, rank: int, n_ranks: int, make_bundler, load, callback, owner: np.ndarray | None = None):
        self.rank, self.n_ranks = rank, n_ranks
        self.owner = partition_landmarks(scene.obs_pt, scene.n_pts, n_ranks) if owner is None else owner
        self.sub, self.pt_ids, self.obs_ids = shard_scene(scene, self.owner, rank, n_ranks)
        self.bundler = make_bundler()
        self._cb = callback
        L = lib()
        L.mage_ba_set_landmark_shard.argtypes = [C.c_void_p, C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p]
        check(L.mage_ba_set_landmark_shard(self.bundler._h, rank, n_ranks, callback, None))
        load(self.bundler, self.sub)

    def StepBundleAdjustment(self, huber_width_per_iteration, max_error_square, outliers: list) -> float:
        own: list = []
        mse = self.bundler.StepBundleAdjustment(huber_width_per_iteration, max_error_square, own)
        outliers.extend(int(self.obs_ids[i]) for i in own)
        return mse

    def GetCurrentLambda(self) -> float:
        return self.bundler.GetCurrentLambda()

    def poses_f64(self) -> np.ndarray:
        return self.bundler.poses_f64()

    def points_into(self, out: np.ndarray) -> None:
        out[self.pt_ids] = self.bundler.points_f64()

    def trace(self):
        return self.bundler.trace()

    def close(self):
        self.bundler.close()


def solve_on_threads(scene: Scene, n_ranks: int, make_bundler, load, calls, prepare=None):
    """Runs all ranks of a landmark-sharded map in this process, one thread each (several handles on one GPU); `calls` is a list of
    (Huber widths, outlier threshold) = one StepBundleAdjustment each.  Returns a dict: per-rank pose arrays, the map-sized point
    array, per call the sorted outlier indices of the map / the mean errors the ranks got / rank 0's iteration trace, the final
    lambdas, the group.  A test vehicle, and a way to split one map's Schur work for a host that drives its GPUs from one process."""
    group = ThreadGroup(n_ranks)
    owner = partition_landmarks(scene.obs_pt, scene.n_pts, n_ranks)
    shards = [ShardedBundler(scene, r, n_ranks, make_bundler, load, group.callback(r), owner) for r in range(n_ranks)]
    if prepare is not None:
        for s in shards:
            prepare(s.bundler)
    outliers = [[[] for _ in calls] for _ in range(n_ranks)]
    mses = [[] for _ in range(n_ranks)]
    traces = [[] for _ in range(n_ranks)]
    errors: list = []

    def run(r):
        try:
            for k, (hubers, thr) in enumerate(calls):
                mses[r].append(shards[r].StepBundleAdjustment(hubers, thr, outliers[r][k]))
                traces[r].append(shards[r].trace())
        except Exception as e:  # noqa: BLE001
            errors.append((r, e))
            group.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(n_ranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    try:
        if errors:
            raise RuntimeError(f"rank {errors[0][0]}: {errors[0][1]}")
        points = np.zeros((scene.n_pts, 3))
        for s in shards:
            s.points_into(points)
        return dict(poses=[s.poses_f64() for s in shards], points=points,
                    outliers=[sorted(i for r in range(n_ranks) for i in outliers[r][k]) for k in range(len(calls))],
                    mse=mses, traces=traces, lambdas=[s.GetCurrentLambda() for s in shards], group=group, owner=owner)
    finally:
        for s in shards:
            s.close()
