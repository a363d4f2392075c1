"""The CLI's --repeat path (gligen_inference.run) called twice in one process, with fork creations counted and the host time of every
generate() call printed: where does the second per-batch 85 ms go that tools/dbg_lanes.py does not see?
   PYTHONPATH=. python tools/dbg_cli.py"""
import time

import torch

import gligen_inference as gi

dev = torch.device("cuda", 0)
gi.device = dev
model, ae, diffusion, cfg = gi.load_synthetic("text")
meta = gi._synthetic_meta("text", 4)
forks = [0]
real_clone = gi._lane_clone


def counting_clone(m):
    forks[0] += 1
    return real_clone(m)


gi._lane_clone = counting_clone
real_generate = gi.generate
log = []


def timed_generate(*a, **k):
    t0 = time.perf_counter()
    r = real_generate(*a, **k)
    log.append((time.perf_counter() - t0) * 1e3)
    return r


gi.generate = timed_generate
gi.save_images = lambda *a, **k: None
for call in range(2):
    log.clear()
    args = dict(batch_size=4, guidance_scale=7.5, negative_prompt="", no_plms=False, folder="/tmp/dbg_cli_out", seed=0, lanes=2, repeat=12, warmup=1 if call == 0 else 0)
    gi.run(meta, args, models=(model, ae, None, diffusion, cfg))
    print(f"call {call}: forks so far {forks[0]}; host ms per generate(): " + " ".join(f"{v:.0f}" for v in log), flush=True)
