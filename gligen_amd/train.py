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

from typing import Dict, Mapping, Optional

import torch

from .dist import GradBuckets


def trainable_names(state_dict: Mapping[str, torch.Tensor]):
    """trainer.py:217-245: 'transformer_blocks' + 'fuser' in the name, or 'position_net'."""
    return [k for k in state_dict if ".fuser." in k or k.startswith("position_net.")]


class TrainStep:
    def __init__(self, engine, cfg: Mapping, state_dict: Mapping[str, torch.Tensor], lr: float = 5e-5, weight_decay: float = 0.0,
                 betas=(0.9, 0.999), eps: float = 1e-8, bucket_mb: float = 128.0, world: Optional[int] = None, checkpoint: bool = True):
        self.engine, self.cfg = engine, dict(cfg)
        dev = engine.device
        self.lr, self.wd, self.betas, self.eps = float(lr), float(weight_decay), tuple(betas), float(eps)
        self.checkpoint = bool(checkpoint)       # activation checkpointing per block (the reference: use_checkpoint=True in every shipped config)
        names = trainable_names(state_dict)
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
        self.steps = 0

    def step(self, batch: Mapping[str, torch.Tensor], fuser_scale: float = 1.0):
        """One iteration: forward, loss, backward, gradient average over the ranks, AdamW. Returns (loss of this rank, eps)."""
        loss, eps, _ = self.engine.unet_train_step(self.cfg, self.params, batch, fuser_scale=fuser_scale, grads=self.gbuf.views, checkpoint=self.checkpoint)
        self.gbuf.all_reduce(average=True)
        self.steps += 1
        for p, g, m, v in zip(self.pbuf.buckets, self.gbuf.buckets, self.m, self.v):
            self.engine.op_adamw_step(p, g, m, v, self.steps, lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.wd)
        return loss, eps

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: v.clone() for k, v in self.params.items()}
