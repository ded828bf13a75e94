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


def init(backend: str, info: RankInfo, device_index: int | None = None, timeout_s: float = 300.0):
    """Initialise torch.distributed when world > 1; returns the module or None.

    The ranks only ever exchange barriers and three scalars, so the control plane must not be what loses a run: the
    rendezvous store is created once, "nccl" (RCCL) is brought up eagerly on this rank's GPU and proven with one
    all-reduce, and if that raises (a node whose RCCL cannot initialise) every rank falls back to "gloo" over the same
    store and says so on stderr.  `init.backend` records what is in use.
    """
    if info.world <= 1:
        return None
    import datetime
    import sys
    import torch
    import torch.distributed as dist
    timeout = datetime.timedelta(seconds=timeout_s)
    # Same rule as torch's env:// rendezvous: under torchrun the agent already hosts the store on MASTER_PORT and every
    # rank is a client; launched by hand, rank 0 hosts it.
    agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
    store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")),
                          info.world, (info.rank == 0) and not agent_store, timeout=timeout, multi_tenant=not agent_store)
    if backend == "nccl":
        dev = info.local_rank if device_index is None else device_index
        try:
            dist.init_process_group(backend="nccl", store=dist.PrefixStore("rccl", store), rank=info.rank, world_size=info.world,
                                    timeout=timeout, device_id=torch.device("cuda", dev))
            probe = torch.ones(1, device=torch.device("cuda", dev))
            dist.all_reduce(probe)
            torch.cuda.synchronize(dev)
            if int(probe.item()) != info.world:
                raise RuntimeError(f"RCCL all-reduce returned {probe.item()} for world {info.world}")
            init.backend = "nccl"
            return dist
        except Exception as e:  # noqa: BLE001 - any RCCL bring-up failure takes the same path
            print(f"[mageslam_amd.dist] rank {info.rank}: RCCL bring-up failed ({type(e).__name__}: {e}); "
                  f"falling back to gloo for barriers and scalar reductions", file=sys.stderr, flush=True)
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:  # noqa: BLE001
                pass
            backend = "gloo"
    dist.init_process_group(backend=backend, store=dist.PrefixStore(backend, store), rank=info.rank, world_size=info.world,
                            timeout=timeout)
    init.backend = backend
    return dist


init.backend = None


def stats_device(device_index: int) -> str:
    """Where the scalar reductions live: on the GPU for RCCL, on the host for gloo."""
    return f"cuda:{device_index}" if init.backend == "nccl" else "cpu"


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
