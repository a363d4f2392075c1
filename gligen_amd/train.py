"""Host side of one optimisation step of the reference's trainer on MI355X.

Reference: trainer.py:353-392 (run_one_step: model(input) on the noised latent, mse_loss(model_output, noise); loss.backward();
opt.step()), :217-245 (the trainable set: every fuser.* parameter and position_net; torch.optim.AdamW over it), :321-322 +
distributed.py:53-62 (DistributedDataParallel: gradients averaged over the ranks).

Everything numeric runs in the library: forward + backward in gl_unet_train_step, the update in gl_op_adamw_step. What lives here is
the bookkeeping the reference leaves to torch.optim / DDP: the trainable parameters and their gradients are views into a few flat
fp32 buffers (gligen_amd.dist.GradBuckets), so the backward writes the gradients where the collective reads them, one
reduce-scatter + all-gather pair per bucket goes over RCCL, and AdamW is one launch per bucket over the flat range.
The step is built for the three discrete grounding tokenizers with gatedSA fusers (what the spatial-map modalities
need); the batch dict carries boxes + masks + positive_embeddings (text), + text_embeddings / image_embeddings / text_masks / image_masks
(text+image), or points + masks (keypoint)."""
from __future__ import annotations

import math
import random
from typing import Callable, Dict, Mapping, Optional, Union

import torch
import torch.distributed as tdist

from .dist import GradBuckets

GROUNDING_KEYS = ("boxes", "masks", "positive_embeddings", "text_embeddings", "image_embeddings", "text_masks", "image_masks", "points")


def warmup_schedule(base_lr: float, warmup_steps: int, total_iters: Optional[int] = None) -> Callable[[int], float]:
    """The reference's LR schedules (trainer.py:262-267, transformers' get_constant_ / get_cosine_schedule_with_warmup):
    linear warm-up over `warmup_steps`, then constant -- or, with total_iters, half a cosine down to 0. step counts from 1
    (the optimiser step about to be taken), i.e. LambdaLR's epoch + 1 at the time opt.step() uses the rate."""
    def lr(step: int) -> float:
        k = step - 1                     # LambdaLR's epoch when this step's update is computed
        if k < warmup_steps:
            return base_lr * k / max(1, warmup_steps)
        if total_iters is None:
            return base_lr
        prog = (k - warmup_steps) / max(1, total_iters - warmup_steps)
        return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
    return lr


def null_grounding(batch: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """GroundingNetInput.get_null_input (grounding_input/*_tokinzer_input.py:30-45): every grounding tensor zeroed."""
    return {k: (torch.zeros_like(v) if k in GROUNDING_KEYS else v) for k, v in batch.items()}


def trainable_names(state_dict: Mapping[str, torch.Tensor]):
    """trainer.py:217-245: 'transformer_blocks' + 'fuser' in the name, or 'position_net'."""
    return [k for k in state_dict if ".fuser." in k or k.startswith("position_net.")]


def _block_order(prefix: str):
    """Module order of a SpatialTransformer's path, whatever order the dict was iterated in: input_blocks.N (numeric N, so
    input_blocks.10 follows input_blocks.2), then middle_block.N, then output_blocks.N -- the order UNetModel.forward walks and
    gl_unet_train_step numbers the blocks in."""
    parts = prefix.split(".")
    stage = {"input_blocks": 0, "middle_block": 1, "output_blocks": 2}.get(parts[0])
    if stage is None:
        raise ValueError(f"trainable fuser outside input_blocks / middle_block / output_blocks: {prefix!r}")
    return (stage,) + tuple(int(p) if p.isdigit() else -1 for p in parts[1:])


def gradient_milestones(names):
    """For every trainable tensor the milestone of gl_unet_train_step behind which its gradient is final (gl_train_wait_grads): a fuser
    tensor's is the number of its SpatialTransformer in MODULE order (input_blocks .., middle_block, output_blocks .., by the numeric
    index in the path -- not the iteration order of the dict handed in: a re-sorted state_dict puts input_blocks.10 before
    input_blocks.2 and would point a bucket at the wrong event), position_net's is the number of SpatialTransformers -- the end of the
    backward. The backward walks the blocks from the last to the first, so milestone j is reached before milestone j - 1."""
    blocks = sorted({k.split(".transformer_blocks.")[0] for k in names if ".fuser." in k}, key=_block_order)
    index = {b: i for i, b in enumerate(blocks)}
    return {k: (index[k.split(".transformer_blocks.")[0]] if ".fuser." in k else len(blocks)) for k in names}


class TrainStep:
    """lr: a float, or a callable step -> rate (warmup_schedule: the reference's warm-up schedulers). drop_prob: the probability with
    which an iteration trains on the null grounding input (UNetModel.forward, openaimodel.py:428: 0.1 while training; 0 here by
    default so that a step is a pure function of its batch) -- drawn from `rng` (random.Random; seed it identically on every rank
    or not at all, as the reference does). With torch.distributed initialised, rank 0's trainable parameters are broadcast once at
    construction, as DistributedDataParallel does (trainer.py:321-322): replicas that start from different state_dicts would
    otherwise drift apart silently.

    overlap (default): the gradient exchange runs UNDER the backward, as DDP's does. The buckets are laid out in the order the
    gradients become final (last SpatialTransformer first, position_net last); after the training step has been enqueued, each
    bucket's reduce-scatter + all-gather and its AdamW update go to a communication stream that waits only for that bucket's
    milestone (gl_train_wait_grads), so bucket 0 is on the wire while the encoder blocks are still in backward; the compute stream
    joins the communication stream at the end of the step. overlap=False: backward, then every collective, then every update, on
    one stream (the round-4 schedule) -- the same numbers bit for bit (same kernels per bucket, same order inside a bucket).

    Arena: the engine's context must hold one block's recompute working set plus the saved block inputs -- 24 GB covers the shipped
    topology at batch 4 x 64 x 64 with checkpoint=True (bench.py / tools/train_bench.py create Engine(arena_gb=24)); the library
    raises 'arena exhausted' (GL_ERR_RUNTIME) rather than spilling when it does not."""

    def __init__(self, engine, cfg: Mapping, state_dict: Mapping[str, torch.Tensor], lr: Union[float, Callable[[int], float]] = 5e-5,
                 weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8, bucket_mb: float = 128.0, world: Optional[int] = None,
                 checkpoint: bool = True, drop_prob: float = 0.0, rng: Optional[random.Random] = None, broadcast: bool = True, overlap: bool = True,
                 cache_frozen: bool = True, exchange_even_alone: bool = False):
        self.engine, self.cfg = engine, dict(cfg)
        # (a world of one rank issues no collective; True sends the buckets through the collectives anyway -- the one-GPU test of the
        # overlapped schedule with RCCL's kernels on the communication stream, tests/test_ops_gpu.py)
        self.exchange_even_alone = bool(exchange_even_alone)
        dev = engine.device
        self.lr = lr if callable(lr) else float(lr)
        self.wd, self.betas, self.eps = float(weight_decay), tuple(betas), float(eps)
        self.drop_prob, self.rng = float(drop_prob), rng or random.Random()
        self.checkpoint = bool(checkpoint)       # activation checkpointing per block (the reference: use_checkpoint=True in every shipped config)
        names = trainable_names(state_dict)
        self.milestone = gradient_milestones(names)
        n_blocks = max(self.milestone.values(), default=0)
        # the engine numbers SpatialTransformers by walking the config; both counts must agree or bucket_ready waits on the wrong events
        n_st = getattr(engine, "count_spatial_transformers", None)
        if callable(n_st) and any(".fuser." in k for k in names) and n_st(self.cfg) != n_blocks:
            raise ValueError(f"TrainStep: {n_blocks} fuser blocks in the state_dict, {n_st(self.cfg)} SpatialTransformers in the config")
        # bucket order = the order in which gradients become final: blocks from the last to the first, position_net at the end
        names = sorted(names, key=lambda k: (self.milestone[k] == n_blocks, -self.milestone[k]))
        shapes = {k: tuple(state_dict[k].shape) for k in names}
        # parameters, gradients and the two AdamW moments share one bucket layout
        self.pbuf = GradBuckets(shapes, bucket_mb, world, device=dev)
        self.gbuf = GradBuckets(shapes, bucket_mb, world, device=dev)
        self.m = [torch.zeros_like(b) for b in self.pbuf.buckets]
        self.v = [torch.zeros_like(b) for b in self.pbuf.buckets]
        self.params: Dict[str, torch.Tensor] = {}
        for k, t in state_dict.items():
            t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            if k in shapes:
                self.pbuf.views[k].copy_(t)
                self.params[k] = self.pbuf.views[k]          # the model's trainable tensors ARE the flat buffers
            else:
                self.params[k] = t
        if broadcast and tdist.is_available() and tdist.is_initialized() and tdist.get_world_size() > 1:
            for b in self.pbuf.buckets:            # the views alias the buckets: one collective per bucket moves every trainable tensor
                tdist.broadcast(b, src=0)
        # a bucket may go out once the LAST of its gradients is written: its lowest block number -- or the end of the backward
        self.bucket_ready = []
        for items in self.gbuf.layout:
            ms = [self.milestone[n] for n, *_ in items]
            self.bucket_ready.append(n_blocks if n_blocks in ms else min(ms))
        self.overlap = bool(overlap) and hasattr(engine, "train_wait_grads")
        self._comm = torch.cuda.Stream(device=dev) if (self.overlap and torch.device(dev).type == "cuda") else None
        self.steps = 0
        # the bf16 operand copies of the frozen parameters are built once and kept on the device (gl_train_weight_cache): this step
        # changes nothing but the tensors it asks gradients for, and load_state_dict drops the copies
        self.cache_frozen = bool(cache_frozen) and hasattr(engine, "train_weight_cache")
        if self.cache_frozen:
            engine.train_weight_cache(False)       # (copies keyed by the addresses of another TrainStep's tensors must not outlive them)
            engine.train_weight_cache(True)

    def lr_at(self, step: int) -> float:
        return float(self.lr(step)) if callable(self.lr) else self.lr

    def step(self, batch: Mapping[str, torch.Tensor], fuser_scale: float = 1.0):
        """One iteration: forward, loss, backward, gradient average over the ranks, AdamW. Returns (loss of this rank, eps)."""
        if self.drop_prob > 0.0 and self.rng.random() < self.drop_prob:      # random drop for guidance (openaimodel.py:428)
            batch = null_grounding(batch)
        kw = dict(use_weight_cache=True) if self.cache_frozen else {}
        loss, eps, _ = self.engine.unet_train_step(self.cfg, self.params, batch, fuser_scale=fuser_scale, grads=self.gbuf.views, checkpoint=self.checkpoint, **kw)
        self.steps += 1
        lr = self.lr_at(self.steps)
        upd = lambda i: self.engine.op_adamw_step(self.pbuf.buckets[i], self.gbuf.buckets[i], self.m[i], self.v[i], self.steps, lr=lr, betas=self.betas,
                                                  eps=self.eps, weight_decay=self.wd)
        nb = len(self.gbuf.buckets)
        alone = self.exchange_even_alone
        if not self.overlap:                        # backward, every collective, every update: one stream
            for i in range(nb):
                self.gbuf.all_reduce_bucket(i, average=True, even_alone=alone)
            for i in range(nb):
                upd(i)
        elif self._comm is None:                    # (a host-side engine: the same per-bucket order without streams)
            for i in range(nb):
                self.engine.train_wait_grads(self.bucket_ready[i], None)
                self.gbuf.all_reduce_bucket(i, average=True, even_alone=alone)
                upd(i)
        else:
            main = torch.cuda.current_stream(self.engine.device)
            for i in range(nb):                     # bucket i: wait for its last gradient only, exchange, update -- all behind the backward
                self.engine.train_wait_grads(self.bucket_ready[i], self._comm)
                with torch.cuda.stream(self._comm):
                    self.gbuf.all_reduce_bucket(i, average=True, even_alone=alone)
                    upd(i)
            main.wait_stream(self._comm)            # the next forward reads the updated parameters
        return loss, eps

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: v.clone() for k, v in self.params.items()}

    def optimizer_state_dict(self) -> Dict[str, object]:
        """What a resumed run needs beside the parameters (the reference's checkpoint carries opt, scheduler and iters,
        trainer.py:472-484): AdamW's two moments per trainable tensor and the step count (which also positions the LR schedule)."""
        m, v = {}, {}
        for items, mb, vb in zip(self.pbuf.layout, self.m, self.v):
            for name, off, n, shape in items:
                m[name] = mb[off:off + n].view(shape).clone()
                v[name] = vb[off:off + n].view(shape).clone()
        return {"steps": int(self.steps), "exp_avg": m, "exp_avg_sq": v}

    def load_optimizer_state_dict(self, state: Mapping[str, object]) -> None:
        for items, mb, vb in zip(self.pbuf.layout, self.m, self.v):
            for name, off, n, shape in items:
                mb[off:off + n].view(shape).copy_(state["exp_avg"][name])
                vb[off:off + n].view(shape).copy_(state["exp_avg_sq"][name])
        self.steps = int(state["steps"])

    def load_state_dict(self, state_dict: Mapping[str, torch.Tensor]) -> None:
        """Parameters back into the flat buffers (trainable) / the frozen set, in place: the views the engine reads stay the same."""
        for k, t in state_dict.items():
            self.params[k].copy_(t.to(device=self.params[k].device, dtype=torch.float32))
        if self.cache_frozen:              # frozen tensors may have changed under the cached operand copies
            self.engine.train_weight_cache(False)
            self.engine.train_weight_cache(True)
