"""Benchmark of the GLIGEN denoising hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: B=4 box+text prompts (8 boxes each) at
512x512, 50 PLMS steps with classifier-free guidance 7.5 (102 UNet forwards per sample, run as 51
[cond ; uncond]-batched evaluations) + AutoencoderKL.decode -> B images (config C2 of BASELINE.json).
Inputs (x_T, CLIP-shaped context, grounding features, seeded random weights of the SD-1.4 GLIGEN
architecture) are resident in HBM before the timed region. N > 1: every rank runs the same batch
size on its own GPU (weak scaling, no data-path collective); value = all images / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_UNET = 1136.88e9     # algorithmic FLOPs per UNetModel.forward per sample, 64x64 latent, Ng = 30
F_VAE_DEC = 2514.52e9  # AutoencoderKL.decode 64x64 -> 512x512 per sample
F_IMG = 102 * F_UNET + F_VAE_DEC
PEAK_BF16 = 2500e12    # dense bf16 MFMA peak, MI355X_MICROARCH.md


def cpu_baseline():
    """The CPU oracle (port of the reference algorithm) on the host cores: one UNet forward and one decode at the
    benchmark's size, extrapolated to 102 forwards + 1 decode per image (a full image is ~15 CPU-minutes)."""
    from gligen_amd import synthetic as syn
    from oracle import gligen_oracle as orc
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import grounding_kwargs, oracle_cfg
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_shapes.json")))
    torch.manual_seed(0)
    sd = {k: torch.randn(s) * 0.02 if len(s) else torch.tensor(0.5) for k, s in shapes["unet_full_text"].items()}
    batch = syn.make_batch("text", 1, n_valid=8)
    inp = dict(x=syn.make_latent(1, 4, 64, 64), timesteps=torch.tensor([501]), context=syn.make_context(1),
               grounding_input=grounding_kwargs("text", batch))
    cfg = oracle_cfg(syn.UNET_CFG, "text")
    with torch.no_grad():
        t0 = time.perf_counter(); orc.unet_forward(sd, cfg, inp); t_unet = time.perf_counter() - t0
        t0 = time.perf_counter(); orc.unet_forward(sd, cfg, inp); t_unet = min(t_unet, time.perf_counter() - t0)
        vsd = {k: torch.randn(s) * 0.02 for k, s in shapes["vae_full"].items()}
        d = syn.VAE_DDCONFIG
        t0 = time.perf_counter()
        orc.vae_decode(vsd, dict(ch_mult=d["ch_mult"], num_res_blocks=d["num_res_blocks"], scale_factor=0.18215), syn.make_latent(1, 4, 64, 64))
        t_dec = time.perf_counter() - t0
    return {"value": 1.0 / (102 * t_unet + t_dec), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/gligen_oracle.py fp32 torch-CPU: 2x unet_forward (B=1, 64x64 latent, Ng=30; best {t_unet:.2f} s) + 1x vae_decode "
                      f"({t_dec:.2f} s), extrapolated to 102 forwards + 1 decode per 512x512 image",
            "host_cpus": os.cpu_count()}


def pmc_traffic(kernel):
    """Fabric-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (tools/gpu_traffic.sh -> profiles/<round>/pmc_traffic.csv; counters cannot be collected from inside a timed run).
    FETCH_SIZE / WRITE_SIZE are KiB summed over the L2 channels; FETCH_SIZE is doubled, the guide's gfx950 correction
    for 16-B-per-lane streaming reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as reported."""
    import csv, glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*", "pmc_traffic.csv")))
    if not files:
        return None, None
    sym = kernel.split(" + ")[0]
    fetch = write = None
    with open(files[-1], newline="") as fh:
        for r in csv.DictReader(fh):
            if r["kernel"] == sym and r["counter"] == "FETCH_SIZE":
                fetch = float(r["mean"]) * 1024 * 2
            if r["kernel"] == sym and r["counter"] == "WRITE_SIZE":
                write = float(r["mean"]) * 1024
    if fetch is None or write is None:
        return None, None
    return fetch + write, (f"{os.path.relpath(files[-1], os.path.dirname(os.path.abspath(__file__)))}: mean over the run's launches of this symbol, "
                           f"2 x FETCH_SIZE ({fetch / 1e6:.1f} MB) + WRITE_SIZE ({write / 1e6:.1f} MB); L2-to-fabric requests, "
                           "Infinity-Cache hits included, so an upper bound on HBM bytes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU per step (BASELINE config C2: 4)")
    ap.add_argument("--plms-steps", type=int, default=50)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--lanes", type=int, default=2,
                    help="batches in flight per GPU: consecutive steps are issued round-robin to this many independent engine "
                         "contexts (own weights copy, arena, hipGraph, HIP stream), so one batch's kernel tails, launch gaps and "
                         "memory-bound kernels overlap the other's MFMA work. 1 = strictly one batch at a time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from gligen_amd import dist as gdist
    rank, local_rank, world = gdist.init_from_env()
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import gligen_inference as gi
    from gligen_amd import synthetic as syn
    from gligen_amd.build import build_native
    if local_rank == 0:
        build_native()       # no-op when the in-tree library is newer than its sources; one rank per node may compile
    gdist.barrier()
    gi.device = dev
    B = args.batch
    # random-init weights of the shipped architecture, generated on the device (fast), same statistics as the test fixture
    L = max(1, min(args.lanes, args.steps))
    lanes = []
    for _ in range(L):
        model, autoencoder, diffusion, cfg = gi.load_synthetic("text", seed=1234, fast=True)
        model.grounding_tokenizer_input = gi.instantiate_from_config(cfg["grounding_tokenizer_input"])
        lanes.append((model, autoencoder, diffusion, torch.cuda.Stream(device=dev)))
    lo, hi = gdist.shard_range(B * world, rank, world)
    batch = {k: v[lo:hi].to(dev) for k, v in syn.make_batch("text", B * world, n_valid=8, seed=0).items()}
    context = syn.make_context(B * world, seed=0)[lo:hi].to(dev)
    uc = syn.make_context(B * world, seed=1)[lo:hi].to(dev)
    x_T = syn.make_latent(B * world, 4, 64, 64, seed=0)[lo:hi].to(dev)
    torch.cuda.synchronize()

    def one_pass(lane):
        model, autoencoder, diffusion, stream = lanes[lane]
        with torch.cuda.stream(stream):
            imgs = gi.generate(model, autoencoder, diffusion, batch, context, uc, steps=args.plms_steps, guidance_scale=7.5,
                               alpha_type=None, starting_noise=x_T.clone(), use_graph=not args.no_graph)
            return autoencoder.engine.to_uint8(imgs)

    for _ in range(max(1, args.warmup)):       # every lane: GEMM autotune (first lane), graph capture, allocator warm-up
        for lane in range(L):
            one_pass(lane)
            torch.cuda.synchronize()
    # UNet evaluation time with the GPU to itself: one more untimed pass on lane 0 alone
    one_pass(0)
    torch.cuda.synchronize()
    unet_ms, first_ms, n_evals = lanes[0][0].engine.sampler_timing()

    gdist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = [one_pass(i % L) for i in range(args.steps)]
    torch.cuda.synchronize(); gdist.barrier()
    elapsed = gdist.max_over_ranks(time.perf_counter() - t0, dev)
    out = outs[-1]
    assert out.shape == (B, 512, 512, 3) and out.dtype == torch.uint8
    assert all(torch.equal(o, out) for o in outs), "lanes disagree on identical inputs"

    # per-kernel profile of one eager [cond ; uncond] evaluation on lane 0 (conditioning is still set from the last pass):
    # HIP events on the launch stream around every launch, aggregated by kernel symbol
    with torch.cuda.stream(lanes[0][3]):
        tt = torch.full((2 * B,), 501, device=dev, dtype=torch.long)
        lanes[0][0].engine.unet_profile(x_T, tt, batch=2 * B)          # warm
        prof = lanes[0][0].engine.unet_profile(x_T, tt, batch=2 * B)
    torch.cuda.synchronize()

    autoencoder = lanes[0][1]
    t0 = time.perf_counter(); d = autoencoder.decode(torch.randn(B, 4, 64, 64, device=dev)); torch.cuda.synchronize()
    dec_ms = (time.perf_counter() - t0) * 1e3
    del d

    if rank == 0:
        n_images = B * world * args.steps
        value = n_images / elapsed
        unet_tflops = 2 * B * F_UNET / (unet_ms * 1e-3) / 1e12
        # dominant kernel = the symbol with the largest total time inside one UNet evaluation; its launches' algorithmic
        # FLOPs / their HIP-event time. profiles/<round>/bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of this
        # same command) lists the same symbol with its average duration over the whole run.
        mm = [p for p in prof if p["flops"] > 0]
        dom = mm[0]
        traffic, traffic_src = pmc_traffic(dom["name"])
        tot_ms = sum(p["ms"] for p in prof)
        roofline = {"bound": "mfma", "kernel": dom["name"], "achieved": dom["flops"] / (dom["ms"] * 1e-3) / 1e12,
                    "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s", "frac": dom["flops"] / (dom["ms"] * 1e-3) / PEAK_BF16,
                    "traffic": traffic, "traffic_src": traffic_src, "launches_per_unet_eval": dom["calls"], "avg_launch_us": dom["ms"] / dom["calls"] * 1e3,
                    "flops_per_launch_avg": dom["flops"] / dom["calls"], "share_of_unet_eval": dom["ms"] / tot_ms,
                    "measured": "HIP events on the launch stream around each launch of one eager UNet evaluation at batch 2B",
                    "unet_eval": {"achieved": unet_tflops, "frac": unet_tflops * 1e12 / PEAK_BF16, "flops_per_launch": 2 * B * F_UNET,
                                  "desc": "whole [cond ; uncond] evaluation (one hipGraph launch) against the same roof"},
                    "whole_image": {"achieved": value / world * F_IMG / 1e12, "frac": value / world * F_IMG / PEAK_BF16},
                    "kernels": [{"kernel": p["name"], "launches": p["calls"], "ms": round(p["ms"], 4),
                                 **({"TFLOP/s": round(p["flops"] / (p["ms"] * 1e-3) / 1e12, 1)} if p["flops"] > 0 else
                                    {"GB/s": round(p["bytes"] / (p["ms"] * 1e-3) / 1e9, 1)})} for p in prof[:12]]}
        line = {
            "metric": "512x512 images/sec @ 50 PLMS steps, box+text (CFG 7.5), UNet step ms",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "C2: box+text, 8 boxes (30 grounding tokens), 512x512, 50 PLMS steps, CFG 7.5, bf16 storage / fp32 accumulate",
                       "images_per_gpu_per_step": B, "plms_steps": args.plms_steps, "unet_evals_per_image": 2 * (args.plms_steps + 1),
                       "hipgraph": not args.no_graph, "batches_in_flight": L, "weights": "seeded random init of the SD-1.4 GLIGEN architecture (966 tensors, 1.07 B params)"},
            "unet_step_ms": unet_ms, "unet_step_desc": f"one [cond ; uncond] UNet evaluation at batch {2 * B} (hipGraph replay, HIP events on the engine stream, "
                                                       f"measured in an untimed pass with one batch in flight)",
            "vae_decode_ms": dec_ms,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    gdist.shutdown()


if __name__ == "__main__":
    main()
