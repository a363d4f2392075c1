"""Why does gligen_inference.generate_stream (the CLI's --repeat) overlap its two lanes less than bench.py's loop? Same process, same
model, three schedules timed back to back: (A) bench.py's loop (per-lane torch streams, outputs converted to uint8 inside the lane),
(B) generate_stream as the CLI calls it (float images kept), (C) generate_stream + uint8 conversion inside the lane.
   PYTHONPATH=. python tools/dbg_lanes.py [n_batches]"""
import copy
import sys
import time

import torch

import gligen_inference as gi
from gligen_amd import synthetic as syn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
gi.device = dev
model, ae, diffusion, cfg = gi.load_synthetic("text", fast=True)
model.grounding_tokenizer_input = gi.instantiate_from_config(cfg["grounding_tokenizer_input"])
B = 4
batch = {k: v.to(dev) for k, v in syn.make_batch("text", B, n_valid=8, seed=0).items()}
ctx, uc = syn.make_context(B, seed=0).to(dev), syn.make_context(B, seed=1).to(dev)
x_T = syn.make_latent(B, 4, 64, 64, seed=0).to(dev)
lanes = [(model, ae, torch.cuda.Stream(device=dev))]
m1, a1 = gi._lane_clone(model), gi._lane_clone(ae)
m1.grounding_tokenizer_input = copy.copy(model.grounding_tokenizer_input)
lanes.append((m1, a1, torch.cuda.Stream(device=dev)))


def one_pass(lane, u8=True):
    m, a, s = lanes[lane]
    with torch.cuda.stream(s):
        img = gi.generate(m, a, diffusion, batch, ctx, uc, steps=50, guidance_scale=7.5, starting_noise=x_T.clone())
        return a.engine.to_uint8(img) if u8 else img


for ln in range(2):
    one_pass(ln); torch.cuda.synchronize()
for label, fn in (("A bench loop, uint8 in lane", lambda i: one_pass(i % 2)), ("A' bench loop, float images kept", lambda i: one_pass(i % 2, False)),
                  ("A1 one lane only", lambda i: one_pass(0))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = [fn(i) for i in range(n)]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label}: {n * B / dt:.3f} images/s", flush=True)
    del outs
noises = [x_T.clone() for _ in range(n)]
kw = dict(steps=50, guidance_scale=7.5)
gi.generate_stream(model, ae, diffusion, batch, ctx, uc, noises[:1], lanes=2, first_lane=0, **kw); torch.cuda.synchronize()
gi.generate_stream(model, ae, diffusion, batch, ctx, uc, noises[:1], lanes=2, first_lane=1, **kw); torch.cuda.synchronize()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = gi.generate_stream(model, ae, diffusion, batch, ctx, uc, noises, lanes=2, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"B generate_stream (its own lane contexts): {n * B / dt:.3f} images/s", flush=True)
    del outs
