"""Bisect the host block seen in tools/dbg_cli.py: model loaded as the CLI loads it (fast=False), generate_stream fed (V1) synthetic device
inputs as tools/dbg_lanes.py builds them, (V2) the batch / context run() prepares from the CLI's meta, (V3) V2 with device-cloned noises,
(V4) V1 with CPU-drawn noises moved to the device as run() does. Host ms per generate() for each.
   PYTHONPATH=. python tools/dbg_cli2.py [fast]"""
import sys
import time

import torch

import gligen_inference as gi
from gligen_amd import synthetic as syn

dev = torch.device("cuda", 0)
gi.device = dev
model, ae, diffusion, cfg = gi.load_synthetic("text", fast=len(sys.argv) > 1)
model.grounding_tokenizer_input = gi.instantiate_from_config(cfg["grounding_tokenizer_input"])
B, n = 4, 6
real_generate = gi.generate
log = []


def timed_generate(*a, **k):
    t0 = time.perf_counter()
    r = real_generate(*a, **k)
    log.append((time.perf_counter() - t0) * 1e3)
    return r


gi.generate = timed_generate
kw = dict(steps=50, guidance_scale=7.5)


def run_variant(label, batch, ctx, uc, noises):
    for ln in range(2):
        gi.generate_stream(model, ae, diffusion, batch, ctx, uc, noises[:1], lanes=2, first_lane=ln, **kw); torch.cuda.synchronize()
    log.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = gi.generate_stream(model, ae, diffusion, batch, ctx, uc, noises, lanes=2, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label}: {n * B / dt:.3f} images/s; host ms per generate(): " + " ".join(f"{v:.0f}" for v in log), flush=True)


x_T = syn.make_latent(B, 4, 64, 64, seed=0).to(dev)
b1 = {k: v.to(dev) for k, v in syn.make_batch("text", B, n_valid=8, seed=0).items()}
c1, u1 = syn.make_context(B, seed=0).to(dev), syn.make_context(B, seed=1).to(dev)
run_variant("V1 synthetic device inputs, device noises", b1, c1, u1, [x_T.clone() for _ in range(n)])
meta = gi._synthetic_meta("text", B)
b2 = gi.prepare_batch(meta, B)
c2, u2 = meta["context"].to(dev)[0:B], meta["uc"].to(dev)[0:B]
run_variant("V2 run()'s batch / context, CPU-drawn noises", b2, c2, u2,
            [torch.randn((B, 4, 64, 64), generator=torch.Generator().manual_seed(r)).to(dev) for r in range(n)])
run_variant("V3 run()'s batch / context, device noises", b2, c2, u2, [x_T.clone() for _ in range(n)])
run_variant("V4 synthetic inputs, CPU-drawn noises", b1, c1, u1,
            [torch.randn((B, 4, 64, 64), generator=torch.Generator().manual_seed(r)).to(dev) for r in range(n)])
print({k: (tuple(v.shape), str(v.device), v.dtype) for k, v in b2.items()})
