"""Multi-GPU plumbing for the bundle-adjustment back-end: one process per GPU, torch.distributed for rendezvous,
barriers and the end-of-run reductions (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests).

What shards: independent sub-maps / LM restarts -- every rank owns whole
BA problems, so the data path has NO collective; ranks only meet at the timing barriers and to combine scalar
statistics (max elapsed time, summed iteration counts, worst RMSE).  The reduced-camera all-reduce variant is
analysed in DESIGN.md section 8 and deliberately not used for the 1k-pose shape (it is xGMI-bound, not compute-bound).
"""
from __future__ import annotations

import os
from dataclasses import dataclass


@dataclass
class RankInfo:
    rank: int
    local_rank: int
    world: int


def rank_info() -> RankInfo:
    return RankInfo(int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def submap_seed(base_seed: int, rank: int) -> int:
    """Seed of the sub-map owned by `rank`: distinct scenes of identical shape (weak scaling)."""
    return base_seed + 0x100 * rank


def assign_submaps(n_submaps: int, rank: int, world: int) -> list[int]:
    """Round-robin ownership of `n_submaps` independent problems (LM restarts, sub-maps, frame batches)."""
    return list(range(rank, n_submaps, world))


def init(backend: str, info: RankInfo):
    """Initialise torch.distributed when world > 1; returns the module or None."""
    if info.world <= 1:
        return None
    import torch
    import torch.distributed as dist
    kw = {}
    if backend == "nccl":
        kw["device_id"] = torch.device("cuda", info.local_rank)
    dist.init_process_group(backend=backend, **kw)
    return dist


def reduce_stats(dist, elapsed_s: float, iterations: int, rmse: float, device: str = "cpu"):
    """(max elapsed over ranks, total iterations, worst RMSE).  Scalars only -- never on the data path."""
    if dist is None:
        return elapsed_s, iterations, rmse
    import torch
    t = torch.tensor([elapsed_s, rmse], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n = torch.tensor([iterations], dtype=torch.int64, device=device)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t[0].item()), int(n[0].item()), float(t[1].item())
