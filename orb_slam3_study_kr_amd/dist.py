"""Multi-GPU plumbing for the throughput run.

Independent local-BA windows shard embarrassingly: window w goes to rank w mod G, every rank
solves its own shard from its own HBM, and there is NO data-path collective.  torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for tests) is used only for the start /
end barriers and the max-over-ranks timing reduction.
"""
from __future__ import annotations

import os
from dataclasses import dataclass


@dataclass
class RankInfo:
    rank: int = 0
    world: int = 1
    local_rank: int = 0
    initialised: bool = False


def shard_indices(n_total: int, rank: int, world: int) -> list[int]:
    """Indices of the units (windows / frame pairs) owned by `rank`: w mod world == rank."""
    return list(range(rank, n_total, world))


def init_from_env(backend: str | None = None) -> RankInfo:
    """Initialise torch.distributed when launched by torch.distributed.run (WORLD_SIZE > 1)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1:
        return RankInfo(0, 1, local_rank, False)
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return RankInfo(rank, world, local_rank, True)


def barrier(info: RankInfo):
    if info.initialised:
        import torch.distributed as dist
        dist.barrier()


def _tensor(info: RankInfo, values):
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    return torch.tensor(values, dtype=torch.float64, device=dev)


def all_reduce_max(info: RankInfo, value: float) -> float:
    if not info.initialised:
        return value
    import torch.distributed as dist
    t = _tensor(info, [value])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_reduce_sum(info: RankInfo, values: list[float]) -> list[float]:
    if not info.initialised:
        return list(values)
    import torch.distributed as dist
    t = _tensor(info, values)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def all_gather_floats(info: RankInfo, value: float) -> list[float]:
    """One float of every rank, in rank order (an all_gather over the process group)."""
    if not info.initialised:
        return [float(value)]
    import torch
    import torch.distributed as dist
    t = _tensor(info, [value])
    out = [torch.zeros_like(t) for _ in range(info.world)]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def finalize(info: RankInfo):
    if info.initialised:
        import torch.distributed as dist
        dist.destroy_process_group()
