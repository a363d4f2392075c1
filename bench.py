"""Benchmark of the GLIGEN denoising hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config C2|C3|C4|C5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment spawns its N ranks itself (re-executes under
torch.distributed.run on 127.0.0.1 with a free port); rank 0 prints the one JSON line either way.

One "step" = one pass of the hot path over one batch of B = 4 prompts per GPU at 512x512: 50 PLMS steps with
classifier-free guidance 7.5 (102 UNet forwards per sample, run as 51 [cond ; uncond]-batched evaluations) +
AutoencoderKL.decode -> B images. --config picks the BASELINE.json configuration (default C2, the one the metric is quoted on):
    C2  box+text, 8 boxes (30 grounding tokens)
    C3  box+text+image grounding (2 x 30 tokens; BASELINE: batch 32 over 8 GPUs = 4 per GPU)
    C4  inpainting box+text: AutoencoderKL.encode of the input image + 9-channel first conv + per-step q_sample blend,
        all inside the timed region
    C5  keypoint grounding (136 tokens; BASELINE: batch 16 over 4 GPUs = 4 per GPU)
Inputs (x_T, CLIP-shaped context, grounding features, seeded random weights of the SD-1.4 GLIGEN architecture) are
resident in HBM before the timed region. N > 1: every rank runs the same batch size on its own GPU (weak scaling, no
data-path collective); value = all images / max-over-ranks time.

Beside the contract's keys the line carries what makes a number comparable across boxes and rounds:
    box_calibration   float4-copy GB/s, global->LDS DMA TB/s, sustained bf16 MFMA TFLOP/s of THIS box (gl_box_calibrate)
    ff_rows_ab        one-lane UNet evaluation ms with the row-local kernels (feed-forward chains and, since round 6, the projection +
                      LayerNorm + q,k,v^T launch) off / forced on / chosen by the engine's on-device timing (the default), and the timed
                      table itself: the same-box A/B of those kernels
    train_step        one iteration of the reference's trainer step (forward + loss + backward + nothing else) of the shipped
                      topology at the same batch and latent, outside the timed region (rank 0, text / text+image / keypoint models)
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOPs (2 per MAC over conv / linear / attention products, reference-faithful op list)
F_UNET = {30: 1136.88e9, 60: 1143.40e9, 136: 1160.32e9}   # per UNetModel.forward per sample at a 64x64 latent, by grounding tokens
F_VAE_DEC = 2514.52e9   # AutoencoderKL.decode 64x64 -> 512x512 per sample
F_VAE_ENC = 1116.66e9   # AutoencoderKL.encode 512x512 -> 64x64 per sample (C4)
F_CONV9 = 12e9 / 102    # the 5 extra first-conv input channels of the inpainting model, per forward
PEAK_BF16 = 2500e12     # dense bf16 MFMA peak, MI355X_MICROARCH.md

CONFIGS = {
    "C2": dict(kind="text", inpaint=False, ng=30, desc="C2: box+text, 8 boxes (30 grounding tokens)"),
    "C3": dict(kind="text_image", inpaint=False, ng=60, desc="C3: box+text+image grounding, 8 boxes (2 x 30 grounding tokens)"),
    "C4": dict(kind="text", inpaint=True, ng=30, desc="C4: inpainting box+text (encode of the input image + 9-channel first conv + per-step blend in the timed region)"),
    "C5": dict(kind="keypoint", inpaint=False, ng=136, desc="C5: keypoint grounding, 2 persons (136 grounding tokens)"),
}


def flops_per_image(cfg, plms_steps=50):
    f = 2 * (plms_steps + 1) * F_UNET[cfg["ng"]] + F_VAE_DEC
    if cfg["inpaint"]:
        f += F_VAE_ENC + 2 * (plms_steps + 1) * F_CONV9
    return f


def cpu_baseline(cfg):
    """The reference's CPU path on the host cores, next to the GPU number: 1 warm-up + 3 timed UNetModel.forward (B=1, 64x64
    latent) + 1 AutoencoderKL.decode, extrapolated to 102 forwards + 1 decode per image (a full image is ~15 CPU-minutes).
    kind "reference": the reference's own modules (oracle/ref_cpu_baseline.py imports them from /root/reference in a
    subprocess; only where that tree is mounted). kind "port": the CPU oracle, oracle/gligen_oracle.py (the GPU box)."""
    script = os.path.join(ROOT, "oracle", "ref_cpu_baseline.py")
    if os.path.isdir("/root/reference/ldm"):
        try:
            r = subprocess.run([sys.executable, script, "--kind", cfg["kind"]], capture_output=True, text=True, timeout=900, cwd="/tmp")
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode == 0 and line:
                return json.loads(line[-1])
        except Exception:
            pass
    from gligen_amd import synthetic as syn
    from oracle import gligen_oracle as orc
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import grounding_kwargs, oracle_cfg
    shapes = json.load(open(os.path.join(ROOT, "tests", "golden", "state_dict_shapes.json")))
    torch.manual_seed(0)
    kind = "text"   # the three tokenizers differ by < 2 % of a forward; the committed shape table is the box+text model's
    sd = {k: torch.randn(s) * 0.02 if len(s) else torch.tensor(0.5) for k, s in shapes["unet_full_text"].items()}
    batch = syn.make_batch(kind, 1, n_valid=8)
    inp = dict(x=syn.make_latent(1, 4, 64, 64), timesteps=torch.tensor([501]), context=syn.make_context(1),
               grounding_input=grounding_kwargs(kind, batch))
    ocfg = oracle_cfg(syn.UNET_CFG, kind)
    ts = []
    with torch.no_grad():
        for i in range(4):   # 1 warm-up + 3
            t0 = time.perf_counter(); orc.unet_forward(sd, ocfg, inp); ts.append(time.perf_counter() - t0)
        t_unet = float(np.mean(ts[1:]))
        vsd = {k: torch.randn(s) * 0.02 for k, s in shapes["vae_full"].items()}
        d = syn.VAE_DDCONFIG
        t0 = time.perf_counter()
        orc.vae_decode(vsd, dict(ch_mult=d["ch_mult"], num_res_blocks=d["num_res_blocks"], scale_factor=0.18215), syn.make_latent(1, 4, 64, 64))
        t_dec = time.perf_counter() - t0
    # /root/reference does not exist on this box, so the stated baseline (the reference's own modules) cannot be timed here:
    # the line says so and carries the number oracle/ref_cpu_baseline.py measured in the build container beside the port's
    ref_bc = None
    try:
        ref_bc = json.load(open(os.path.join(ROOT, "oracle", "ref_cpu_baseline_build_container.json")))
    except Exception:
        pass
    return {"reference_build_container": ref_bc,
            "note": "/root/reference is absent on this host: 'value' times the CPU oracle (a port); 'reference_build_container' is the "
                    "reference's own modules timed by oracle/ref_cpu_baseline.py in the build container (other host CPU, see its 'cores')",
            "value": 1.0 / (102 * t_unet + t_dec), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/gligen_oracle.py fp32 torch-CPU: 1 warm-up + 3 timed unet_forward (B=1, 64x64 latent, Ng=30; mean {t_unet:.2f} s) "
                      f"+ 1x vae_decode ({t_dec:.2f} s), extrapolated to 102 forwards + 1 decode per 512x512 image",
            "host_cpus": os.cpu_count()}


def pmc_traffic(kernel):
    """Fabric-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (tools/gpu_traffic.sh -> profiles/<round>/pmc_traffic.csv; counters cannot be collected from inside a timed run).
    FETCH_SIZE / WRITE_SIZE are KiB summed over the L2 channels; FETCH_SIZE is doubled, the guide's gfx950 correction
    for 16-B-per-lane streaming reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as reported."""
    import csv
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.csv")))   # rN_final sorts last: the newest round's passes
    if not files:
        return None, None
    sym = kernel.split(" + ")[0].split(" [")[0]
    fetch = write = None
    with open(files[-1], newline="") as fh:
        for r in csv.DictReader(fh):
            if r["kernel"] == sym and r["counter"] == "FETCH_SIZE":
                fetch = float(r["mean"]) * 1024 * 2
            if r["kernel"] == sym and r["counter"] == "WRITE_SIZE":
                write = float(r["mean"]) * 1024
    if fetch is None or write is None:
        return None, None
    return fetch + write, (f"{os.path.relpath(files[-1], ROOT)}: mean over the run's launches of this symbol, "
                           f"2 x FETCH_SIZE ({fetch / 1e6:.1f} MB) + WRITE_SIZE ({write / 1e6:.1f} MB); L2-to-fabric requests, "
                           "Infinity-Cache hits included, so an upper bound on HBM bytes; per-problem rows: pmc_traffic_per_problem.csv beside it")


def train_step_line(model, kind, B, dev):
    """One training iteration (gl_unet_train_step: forward + mse loss + backward for the reference's trainable set, activation
    checkpointing on, trainer.py:353-392) of the benchmark's own model at the benchmark's batch and latent, after one untimed
    warm-up iteration (tile tuning). tflops: 3.2 x the forward's algorithmic FLOPs (forward + recomputed forward + input-gradient
    backward of the frozen layers + weight gradients of the fusers only) / time, against the same bf16 roof."""
    import time as _t
    from gligen_amd import synthetic as syn
    from gligen_amd.engine import Engine
    try:
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        if any(v.dtype != torch.float32 or not v.is_cuda for v in sd.values()):
            sd = {k: v.float().to(dev).contiguous() for k, v in sd.items()}
        b = syn.make_batch(kind, B, n_valid=8, seed=5)
        if kind == "keypoint":
            b["boxes"] = None
        ts = torch.tensor([981, 441, 300, 77, 650, 12, 850, 505][:B] if B <= 8 else list(range(1, 1000, 1000 // B))[:B]).float()
        batch = dict(x=syn.make_latent(B, 4, 64, 64, seed=6), timesteps=ts, context=syn.make_context(B, seed=6), masks=b["masks"],
                     target=syn.make_latent(B, 4, 64, 64, seed=7))
        if kind != "keypoint":
            batch["boxes"] = b["boxes"]
        if kind == "keypoint":
            batch["points"] = b["points"]
        elif kind == "text_image":
            batch.update(text_embeddings=b["text_embeddings"], text_masks=b["text_masks"], image_masks=b["image_masks"], image_embeddings=b["image_embeddings"])
        else:
            batch["positive_embeddings"] = b["text_embeddings"]
        grads = {k: torch.zeros_like(v) for k, v in sd.items() if ".fuser." in k or k.startswith("position_net.")}
        cfg = dict(syn.UNET_CFG, grounding_tokenizer=syn.GROUNDING_TOKENIZERS[kind])
        eng = Engine(dev, arena_gb=24.0)
        eng.train_weight_cache(True)      # as gligen_amd.train.TrainStep runs it: the frozen parameters' bf16 operand copies are built once
        kw = dict(grads=grads, checkpoint=True, use_weight_cache=True)
        eng.unet_train_step(cfg, sd, batch, **kw)
        torch.cuda.synchronize()
        t0 = _t.perf_counter()
        loss, _, _ = eng.unet_train_step(cfg, sd, batch, **kw)
        torch.cuda.synchronize()
        dt = _t.perf_counter() - t0
        cache_gb = eng.train_weight_cache(True) / 2 ** 30
        eng.train_weight_cache(False)
        ng = {"text": 30, "text_image": 60, "keypoint": 136}[kind]
        tf = 3.2 * B * F_UNET[ng] / dt / 1e12
        return {"ms": round(dt * 1e3, 1), "B": B, "latent": 64, "checkpoint": True, "loss": float(loss), "tflops": round(tf, 1), "frac": round(tf * 1e12 / PEAK_BF16, 4),
                "trainable_values": sum(int(g.numel()) for g in grads.values()), "arena_high_water_gb": round(eng.arena_high_water() / 2 ** 30, 2),
                "weight_cache_gb": round(cache_gb, 2),
                "desc": "gl_unet_train_step of the shipped topology: forward + mse_loss + backward (fuser + position_net gradients), fp32 activations, "
                        "three-pass bf16 MFMA products, the frozen parameters' operand copies cached across iterations (gl_train_weight_cache); "
                        "gradient exchange and AdamW not included (1 GPU)"}
    except Exception as e:   # a measurement aid must not take the bench line down
        return {"error": f"{type(e).__name__}: {e}"[:300]}


class ClockSampler(threading.Thread):
    """Shader / memory clock of this rank's GPU while the timed region runs (sysfs pp_dpm_sclk / pp_dpm_mclk, the level marked
    '*'), every 0.25 s: box-to-box spread of the same commit was +-12 % in round 1 and needs a clock next to every number."""

    def __init__(self, index):
        super().__init__(daemon=True)
        self.files = {}
        base = self._sysfs_of(index)
        if base:
            self.files = {"sclk_mhz": os.path.join(base, "pp_dpm_sclk"), "mclk_mhz": os.path.join(base, "pp_dpm_mclk")}
        self.samples = {k: [] for k in self.files}
        self.stop_flag = threading.Event()

    @staticmethod
    def _sysfs_of(index):
        """sysfs directory of HIP device `index`, by PCI bus id (the node's other GPUs are in /sys too, idle at ~130 MHz)."""
        try:
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(index)) == 0:
                d = os.path.join("/sys/bus/pci/devices", buf.value.decode().lower())
                if os.path.exists(os.path.join(d, "pp_dpm_sclk")):
                    return d
        except Exception:
            pass
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
        return os.path.dirname(cards[0]) if len(cards) == 1 else None

    @staticmethod
    def _current(path):
        try:
            for line in open(path):
                if "*" in line:
                    return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except Exception:
            return None
        return None

    def run(self):
        while not self.stop_flag.is_set():
            for k, p in self.files.items():
                v = self._current(p)
                if v is not None:
                    self.samples[k].append(v)
            self.stop_flag.wait(0.25)

    def summary(self):
        self.stop_flag.set()
        out = {}
        for k, v in self.samples.items():
            if v:
                out[k] = {"min": min(v), "mean": round(sum(v) / len(v), 1), "max": max(v), "samples": len(v)}
        return out or None


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` outside a launcher: re-execute under torch.distributed.run, one process per GPU, rendezvous
    on 127.0.0.1 (the container hostname may not resolve). The children's stdout / stderr pass through (rank 0 prints the
    JSON line); the exit code is the launcher's."""
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def stub_lanes(L, B):
    """--stub: the measurement protocol without the engine (CPU test of the N-rank launch path, tests/test_dist_cpu.py):
    a 'pass' sleeps 20 ms and returns a deterministic uint8 tensor of the real shape."""
    def one_pass(lane):
        time.sleep(0.02)
        return torch.zeros((B, 512, 512, 3), dtype=torch.uint8)
    return one_pass


def stub_main(args, cfg, gdist, rank, world, dist_world, dist_backend):
    B = args.batch
    one_pass = stub_lanes(1, B)
    for _ in range(max(1, args.warmup)):
        one_pass(0)
    gdist.barrier()
    t0 = time.perf_counter()
    outs = [one_pass(0) for _ in range(args.steps)]
    gdist.barrier()
    elapsed = gdist.max_over_ranks(time.perf_counter() - t0)
    assert outs[-1].shape == (B, 512, 512, 3)
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": B * world * args.steps / elapsed, "unit": "images/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "stub", "config": {"workload": "stub: " + cfg["desc"]},
                          "collective_world_size": dist_world, "collective_backend": dist_backend}), flush=True)
    gdist.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)       # (a multiple of the batches in flight: no lane idles through the last step)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS), help="BASELINE.json configuration (default: C2, the one the metric is quoted on)")
    ap.add_argument("--batch", type=int, default=4, help="images per GPU per step (BASELINE C2: 4; C3: 32 / 8 GPUs; C5: 16 / 4 GPUs)")
    ap.add_argument("--plms-steps", type=int, default=50)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--lanes", type=int, default=3,
                    help="batches in flight per GPU: consecutive steps are issued round-robin to this many execution contexts (engine "
                         "forks: ONE set of packed weights, own arena, hipGraph, HIP stream each), so one batch's kernel tails, launch "
                         "gaps and memory-bound kernels overlap the other's MFMA work. 1 = strictly one batch at a time; the line "
                         "carries that number too (value_one_lane). Default 3 (round 6, same box, the driver's step counts: 5.73 / 6.50 / "
                         "6.64 / 6.38 images/s at 1 / 2 / 3 / 4, profiles/r6/lanes_sweep.txt)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-step", action="store_true", help="skip the training-iteration line (train_step)")
    ap.add_argument("--no-ff-ab", action="store_true", help="skip the same-box A/B of the row-local kernels (ff_rows_ab)")
    ap.add_argument("--alpha-type", default=None,
                    help="gate schedule 'on,decay,off' (fractions of the steps), e.g. 0.3,0,0.7 as in the reference's demo prompts; default: "
                         "None = fusers on at every step, the configuration the metric is quoted on (and the one with the most work)")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)   # protocol-only run without a GPU (CPU test of the launch path)
    args = ap.parse_args()
    alpha_type = [float(v) for v in args.alpha_type.split(",")] if args.alpha_type else None
    cfg = CONFIGS[args.config]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    from gligen_amd import dist as gdist
    rank, local_rank, world = gdist.init_from_env(backend="gloo" if args.stub else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node and --gpus must agree")
    dist_world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    dist_backend = torch.distributed.get_backend() if torch.distributed.is_initialized() else None
    if args.stub:
        return stub_main(args, cfg, gdist, rank, world, dist_world, dist_backend)
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
    kind = cfg["kind"]
    # random-init weights of the shipped architecture, generated on the device (fast), same statistics as the test fixture
    L = max(1, min(args.lanes, args.steps))
    model, autoencoder, diffusion, mcfg = gi.load_synthetic(kind, inpaint=cfg["inpaint"], seed=1234, fast=True)
    model.grounding_tokenizer_input = gi.instantiate_from_config(mcfg["grounding_tokenizer_input"])
    # (developer A/B: GL_LANE_PRIO="-1,0,0" gives the lanes' streams HIP priorities; measured neutral-to-worse, profiles/r6/lanes_sweep.txt)
    prio = [int(v) for v in os.environ.get("GL_LANE_PRIO", "").split(",") if v.strip()] if os.environ.get("GL_DEV_SWITCHES") else []
    lane_stream = lambda i: torch.cuda.Stream(device=dev, priority=prio[i]) if i < len(prio) else torch.cuda.Stream(device=dev)
    lanes = [(model, autoencoder, diffusion, lane_stream(0))]
    for i in range(1, L):    # further lanes: forks of the first lane's engines (shared packed weights, gl_ctx_fork), own tokenizer-input state
        import copy
        m, ae = gi._lane_clone(model), gi._lane_clone(autoencoder)
        m.grounding_tokenizer_input = copy.copy(model.grounding_tokenizer_input)
        lanes.append((m, ae, diffusion, lane_stream(i)))
    lo, hi = gdist.shard_range(B * world, rank, world)
    batch = {k: v[lo:hi].to(dev) for k, v in syn.make_batch(kind, B * world, n_valid=8, seed=0).items()}
    context = syn.make_context(B * world, seed=0)[lo:hi].to(dev)
    uc = syn.make_context(B * world, seed=1)[lo:hi].to(dev)
    x_T = syn.make_latent(B * world, 4, 64, 64, seed=0)[lo:hi].to(dev)
    image = mask = None
    if cfg["inpaint"]:   # one input image per sample in [-1, 1], mask from the boxes (reference gligen_inference.py:396-407)
        image = (torch.rand(B * world, 3, 512, 512, generator=torch.Generator().manual_seed(8)) * 2 - 1)[lo:hi].to(dev)
        mask = gi.draw_masks_from_boxes(batch["boxes"].cpu(), 64).to(dev)
    if alpha_type is not None and not cfg["inpaint"]:
        # restore_first_conv_from_SD reads a cwd-relative file (reference openaimodel.py:404): a seeded stand-in of the same shapes
        import tempfile
        os.chdir(tempfile.mkdtemp())
        torch.save(syn.sd_first_conv_state(), "SD_input_conv_weight_bias.pth")
    torch.cuda.synchronize()

    def one_pass(lane):
        model, autoencoder, diffusion, stream = lanes[lane]
        with torch.cuda.stream(stream):
            z0 = autoencoder.encode(image) if cfg["inpaint"] else None
            imgs = gi.generate(model, autoencoder, diffusion, batch, context, uc, steps=args.plms_steps, guidance_scale=7.5,
                               alpha_type=alpha_type, starting_noise=x_T.clone(), use_graph=not args.no_graph, inpainting_mask=mask, z0=z0)
            return autoencoder.engine.to_uint8(imgs)

    for _ in range(max(1, args.warmup)):       # every lane: GEMM autotune (first lane), graph capture, allocator warm-up
        for lane in range(L):
            one_pass(lane)
            torch.cuda.synchronize()
    # same-box A/B of the row-local feed-forward kernel, one lane, untimed passes: never / wherever it exists / the engine's own
    # on-device timing (the default, restored last; a policy change drops the captured graphs, so every lane re-captures below)
    eng0 = lanes[0][0].engine
    ff_ab = None
    if not args.no_ff_ab:
        ff_ab = {}
        for name, mode in (("off", 0), ("forced", 1)):
            eng0.set_ff_rows_policy(mode)
            one_pass(0)
            torch.cuda.synchronize()
            ff_ab["unet_step_ms_" + name] = round(eng0.sampler_timing()[0], 4)
        eng0.set_ff_rows_policy(-1)
        for lane in range(L):
            one_pass(lane)
            torch.cuda.synchronize()
    # UNet evaluation time with the GPU to itself: one more untimed pass on lane 0 alone
    one_pass(0)
    torch.cuda.synchronize()
    unet_ms, first_ms, n_evals = eng0.sampler_timing()
    if ff_ab is not None:
        ff_ab["unet_step_ms_timed_choice"] = round(unet_ms, 4)
        ff_ab["timed_table"] = eng0.ff_rows_policy_report()

    clocks = ClockSampler(local_rank)
    gdist.barrier(); torch.cuda.synchronize()
    clocks.start()
    t0 = time.perf_counter()
    outs = [one_pass(i % L) for i in range(args.steps)]
    torch.cuda.synchronize(); gdist.barrier()
    elapsed = gdist.max_over_ranks(time.perf_counter() - t0, dev)
    clk = clocks.summary()
    # the strict BASELINE C2 reading -- one batch at a time on one context -- over the same number of steps, outside the timed
    # region above (what gligen_inference.run() delivers below 8 samples per GPU)
    elapsed_one = None
    if L > 1:
        gdist.barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            one_pass(0)
        torch.cuda.synchronize(); gdist.barrier()
        elapsed_one = gdist.max_over_ranks(time.perf_counter() - t1, dev)
    out = outs[-1]
    assert out.shape == (B, 512, 512, 3) and out.dtype == torch.uint8
    if not cfg["inpaint"]:   # (inpainting draws fresh q_sample / posterior noise every pass, as the reference does)
        assert all(torch.equal(o, out) for o in outs), "lanes disagree on identical inputs"

    # Per-prompt work the timed passes above do not repeat: they hand the engine the SAME context / grounding tensors every pass, so
    # UNetModel.set_conditioning (reference openaimodel.py:196-208) finds its cache and skips position_net, the 32 cross-attention K / V
    # projections and the 16 fuser K / V fills. One more pass on FRESH copies of the conditioning tensors (a new prompt), HIP events around
    # the engine's set_cond: what "x_T -> image" costs beyond the timed number when every batch is a new prompt.
    set_cond = {"calls": 0, "ms": 0.0}
    real_set_cond = type(eng0).set_cond
    cond_events = []

    def timed_set_cond(self, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real_set_cond(self, *a, **k)
        e1.record()
        cond_events.append((e0, e1))
        return r
    type(eng0).set_cond = timed_set_cond
    try:
        m0, ae0, d0, st0 = lanes[0]
        with torch.cuda.stream(st0):
            fresh_batch = {k: v.clone() for k, v in batch.items()}
            z0f = ae0.encode(image) if cfg["inpaint"] else None
            gi.generate(m0, ae0, d0, fresh_batch, context.clone(), uc.clone(), steps=args.plms_steps, guidance_scale=7.5, alpha_type=alpha_type,
                        starting_noise=x_T.clone(), use_graph=not args.no_graph, inpainting_mask=mask, z0=z0f)
        torch.cuda.synchronize()
    finally:
        type(eng0).set_cond = real_set_cond
    set_cond["calls"] = len(cond_events)
    set_cond["ms"] = round(sum(a.elapsed_time(b) for a, b in cond_events), 4)

    # per-kernel profile of one eager [cond ; uncond] evaluation on lane 0 (conditioning is still set from the last pass):
    # HIP events on the launch stream around every launch, aggregated by kernel symbol
    with torch.cuda.stream(lanes[0][3]):
        tt = torch.full((2 * B,), 501, device=dev, dtype=torch.long)
        extra = None
        if cfg["inpaint"]:
            z0 = lanes[0][1].encode(image)
            extra = torch.cat([z0 * mask, mask], dim=1)
        lanes[0][0].engine.unet_profile(x_T, tt, extra, batch=2 * B)          # warm
        n0 = lanes[0][0].engine.launch_count()
        prof = lanes[0][0].engine.unet_profile(x_T, tt, extra, batch=2 * B)
        launches_per_eval = lanes[0][0].engine.launch_count() - n0   # kernel launches of one [cond ; uncond] evaluation (gl_launch_count)
    torch.cuda.synchronize()

    autoencoder = lanes[0][1]
    t0 = time.perf_counter(); d = autoencoder.decode(torch.randn(B, 4, 64, 64, device=dev)); torch.cuda.synchronize()
    dec_ms = (time.perf_counter() - t0) * 1e3
    del d

    mem_line = {"arena_high_water_gb": round(lanes[0][0].engine.arena_high_water() / 2 ** 30, 3),   # activation arena of one execution context (gl_arena_high_water)
                # footprint of this rank: packed weights + slabs of every context (forks share the weights: theirs are slabs only) and the
                # device memory committed behind the arenas (UNet + VAE contexts of every lane)
                "per_gpu_weight_bytes": sum(e.memory()["own_bytes"] for ln in lanes for e in (ln[0].engine, ln[1].engine)),
                "arena_reserved_bytes": sum(e.memory()["arena_bytes"] for ln in lanes for e in (ln[0].engine, ln[1].engine))}
    box = None
    train = None
    if rank == 0:
        try:
            box = eng0.box_calibrate()     # (after the memory numbers above: it borrows 1 GiB of lane 0's arena)
        except Exception as e:
            box = {"error": str(e)[:200]}
        if not args.no_train_step and not cfg["inpaint"] and alpha_type is None and world == 1:
            train = train_step_line(lanes[0][0], kind, B, dev)

    if rank == 0:
        n_images = B * world * args.steps
        value = n_images / elapsed
        f_unet = F_UNET[cfg["ng"]] + (F_CONV9 if cfg["inpaint"] else 0.0)
        f_img = flops_per_image(cfg, args.plms_steps)
        unet_tflops = 2 * B * f_unet / (unet_ms * 1e-3) / 1e12
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
                    "unet_eval": {"achieved": unet_tflops, "frac": unet_tflops * 1e12 / PEAK_BF16, "flops_per_launch": 2 * B * f_unet,
                                  "desc": "whole [cond ; uncond] evaluation (one hipGraph launch) against the same roof"},
                    "whole_image": {"achieved": value / world * f_img / 1e12, "frac": value / world * f_img / PEAK_BF16, "flops_per_image": f_img},
                    "eager_sum_ms": round(tot_ms, 3),
                    "kernels": [{"kernel": p["name"], "launches": p["calls"], "ms": round(p["ms"], 4),
                                 **({"TFLOP/s": round(p["flops"] / (p["ms"] * 1e-3) / 1e12, 1)} if p["flops"] > 0 else
                                    {"GB/s": round(p["bytes"] / (p["ms"] * 1e-3) / 1e9, 1), "frac_of_6.3TB/s": round(p["bytes"] / (p["ms"] * 1e-3) / 6.3e12, 3)})}
                                for p in prof[:14]]}
        line = {
            "metric": "512x512 images/sec @ 50 PLMS steps, box+text (CFG 7.5), UNet step ms",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": cfg["desc"] + ", 512x512, 50 PLMS steps, CFG 7.5, bf16 storage / fp32 accumulate", "baseline_config": args.config,
                       "images_per_gpu_per_step": B, "plms_steps": args.plms_steps, "unet_evals_per_image": 2 * (args.plms_steps + 1),
                       "grounding_tokens": cfg["ng"], "alpha_type": alpha_type or [1, 0, 0], "hipgraph": not args.no_graph, "batches_in_flight": L,
                       "weights": "seeded random init of the SD-1.4 GLIGEN architecture (966 tensors, 1.07 B params)"},
            "unet_step_ms": unet_ms, "unet_step_desc": f"one [cond ; uncond] UNet evaluation at batch {2 * B} (hipGraph replay, HIP events on the engine stream, "
                                                       f"measured in an untimed pass with one batch in flight)",
            "vae_decode_ms": dec_ms,
            "set_cond_ms": set_cond["ms"], "set_cond_desc": f"per-PROMPT work outside the timed passes (they reuse one prompt's conditioning cache): position_net, "
                                                            f"32 cross-attention K/V, 16 fuser K/V for a batch of {2 * B}; {set_cond['calls']} call(s), HIP events, "
                                                            f"= {set_cond['ms'] / max(elapsed / args.steps * 1e3, 1e-9) * 100:.2f} % of ms_per_step if every batch were a new prompt",
            "launches_per_unet_eval": launches_per_eval,
            **mem_line,
            "box_calibration": box,
            "ff_rows_ab": ff_ab,
            "train_step": train,
            "value_one_lane": (B * world * args.steps / elapsed_one) if elapsed_one else value,
            "collective_world_size": dist_world, "collective_backend": dist_backend,   # the RCCL world the barrier / max-over-ranks ran in
            "gpu_clocks": clk,
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:   # rank 0's host cores, the 1-GPU run only (the N > 1 runs of a scaling sweep repeat nothing)
            line["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(line), flush=True)
    gdist.shutdown()


if __name__ == "__main__":
    main()
