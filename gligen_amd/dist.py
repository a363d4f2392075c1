"""One process per GPU. The denoising path shards by sample (independent trajectories, no exchange
step — SURVEY.md §8e), so the only collectives are the benchmark's barrier and max-over-ranks clock."""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """torch.distributed over RCCL ('nccl' on ROCm) when WORLD_SIZE > 1; gloo on CPU-only hosts."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of `total` samples owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return value
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return value
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def shutdown() -> None:
    """Tear the process group down (after a final barrier) so that ranks leave together and RCCL exits quietly."""
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
