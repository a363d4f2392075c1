"""One process per GPU. The denoising path shards by sample (independent trajectories, no exchange
step — SURVEY.md §8e), so the only collectives there are the benchmark's barrier and max-over-ranks clock.
The training path has one real exchange: the gradients of the trainable set, GradBuckets below."""
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


def broadcast_int(value: int, src: int = 0, device=None) -> int:
    """`value` of rank `src` on every rank (one run-wide random seed when the caller gave none: the ranks must slice ONE draw)."""
    if not dist.is_initialized():
        return int(value)
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    t = torch.tensor([int(value)], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=src)
    return int(t.item())


def shutdown() -> None:
    """Tear the process group down (after a final barrier) so that ranks leave together and RCCL exits quietly."""
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


class GradBuckets:
    """Gradient exchange of the training path (reference trainer.py:321-322: DistributedDataParallel over the trainable set --
    16 fusers + position_net, ~209 M fp32 values = 836 MB per step; distributed.py:53-62). The gradients live IN a few large flat
    buffers from the start -- `views[name]` is the tensor the backward kernels write (gl_op_block_train takes its pointer), so
    there is no pack / unpack pass -- and every buffer is exchanged as one collective: xGMI is point-to-point (7 links x ~153 GB/s
    per GPU), a ring collective is per-link bound, and few large transfers amortise its latency where DDP's default 25 MB buckets
    would pay it 34 times. On RCCL each bucket is a reduce-scatter + all-gather pair (the halves of a ring all-reduce, so an
    optimizer that is sharded over the ranks later only drops the second half); gloo (CPU tests) uses all_reduce.
    A tensor is never split across buckets; bucket lengths are padded to a multiple of the world size."""

    def __init__(self, named_shapes, bucket_mb: float = 128.0, world: int | None = None, device="cpu", dtype=torch.float32):
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        limit = max(1, int(bucket_mb * (1 << 20)) // torch.empty((), dtype=dtype).element_size())
        self.layout = []          # per bucket: [(name, offset, numel, shape)]
        cur, used = [], 0
        for name, shape in named_shapes.items():
            n = 1
            for d in shape:
                n *= int(d)
            if cur and used + n > limit:
                self.layout.append(cur)
                cur, used = [], 0
            cur.append((name, used, n, tuple(int(d) for d in shape)))
            used += n
        if cur:
            self.layout.append(cur)
        self.buckets, self.views = [], {}
        for items in self.layout:
            total = items[-1][1] + items[-1][2]
            padded = (total + self.world - 1) // self.world * self.world
            buf = torch.zeros(padded, dtype=dtype, device=device)
            self.buckets.append(buf)
            for name, off, n, shape in items:
                self.views[name] = buf[off:off + n].view(shape)

    def zero_(self):
        for b in self.buckets:
            b.zero_()

    def all_reduce_bucket(self, i: int, average: bool = True, even_alone: bool = False) -> int:
        """Sum (or mean) of bucket i over the ranks, in place, on the current stream. Returns the number of collectives issued.
        A single rank has nothing to exchange and issues none, unless `even_alone` asks for the collectives anyway (the one-GPU
        test of the RCCL branch: a world of one runs the same reduce-scatter + all-gather pair and must leave the bucket as it was)."""
        if not dist.is_initialized() or (self.world == 1 and not even_alone):
            return 0
        buf = self.buckets[i]
        if dist.get_backend() == "nccl":
            shard = buf.numel() // self.world
            mine = torch.empty(shard, dtype=buf.dtype, device=buf.device)
            dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.SUM)
            if average:
                mine.div_(self.world)
            dist.all_gather_into_tensor(buf, mine)
            return 2
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        if average:
            buf.div_(self.world)
        return 1

    def all_reduce(self, average: bool = True):
        """Sum (or mean) of every bucket over the ranks, in place. Returns the number of collectives issued."""
        return sum(self.all_reduce_bucket(i, average) for i in range(len(self.buckets)))
