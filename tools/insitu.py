#!/usr/bin/env python
"""In-situ per-problem GEMM/conv times of one eager UNet evaluation (GL_PROF_SHAPES=1), to set against kbench's
isolated per-shape table: which problems pay for cold weights / cold L2 inside the real forward."""
import os, sys
os.environ["GL_DEV_SWITCHES"] = "1"   # the library reads developer switches only with this set
os.environ["GL_PROF_SHAPES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gligen_inference as gi
from gligen_amd import synthetic as syn

dev = torch.device("cuda", 0)
gi.device = dev
B = int(os.environ.get("INSITU_B", "4"))      # images per batch (the benchmark: 4 = 8 samples per evaluation)
model, ae, diffusion, cfg = gi.load_synthetic("text", seed=1234, fast=True)
model.grounding_tokenizer_input = gi.instantiate_from_config(cfg["grounding_tokenizer_input"])
batch = {k: v.to(dev) for k, v in syn.make_batch("text", B, n_valid=8, seed=0).items()}
context = syn.make_context(B, seed=0).to(dev)
uc = syn.make_context(B, seed=1).to(dev)
x_T = syn.make_latent(B, 4, 64, 64, seed=0).to(dev)
gi.generate(model, ae, diffusion, batch, context, uc, steps=2, guidance_scale=7.5, alpha_type=None, starting_noise=x_T.clone())
tt = torch.full((2 * B,), 501, device=dev, dtype=torch.long)
acc = {}
R = 5
for r in range(R + 1):
    prof = model.engine.unet_profile(x_T, tt, batch=2 * B)
    if r == 0:
        continue
    for p in prof:
        a = acc.setdefault(p["name"], [p["calls"], 0.0, p["flops"]])
        a[1] += p["ms"] / R
tot = sum(a[1] for a in acc.values())
print(f"in-situ total {tot:.3f} ms (event-timed eager launches, mean of {R})")
for name, (calls, ms, flops) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("INSITU_ROWS", "70"))]:
    tf = f"{flops / (ms * 1e-3) / 1e12:7.1f} TF/s" if flops > 0 else ""
    print(f"{name:78s} {calls:4d} x {ms / calls * 1e3:8.1f} us = {ms:7.3f} ms {tf}")
