"""World-size-2 (gloo, CPU) coverage of the multi-GPU path: the denoising path shards by sample with no
data-path collective, so what must hold is (1) the shards partition the global batch and
reproduce the single-process inputs exactly, (2) the bench's barrier / max-over-ranks clock works."""
import os
import socket
import subprocess
import sys
import textwrap

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, {root!r})
    import torch
    from gligen_amd import dist as gdist, synthetic as syn
    rank, local_rank, world = gdist.init_from_env(backend="gloo")
    assert world == 2
    B = 3  # per-rank batch as bench.py builds it
    lo, hi = gdist.shard_range(B * world, rank, world)
    x = syn.make_latent(B * world, 4, 8, 8, seed=0)[lo:hi]
    ctx = syn.make_context(B * world, seed=0)[lo:hi]
    batch = {{k: v[lo:hi] for k, v in syn.make_batch("text", B * world, n_valid=8, seed=0).items()}}
    gdist.barrier()
    t = gdist.max_over_ranks(1.0 + rank)
    s = gdist.sum_over_ranks(float(hi - lo))
    out = dict(rank=rank, lo=lo, hi=hi, t=t, s=s, x=float(x.double().sum()), ctx=float(ctx.double().sum()),
               boxes=float(batch["boxes"].double().sum()), n=int(x.shape[0]))
    print("RESULT " + json.dumps(out))
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_partitions():
    from gligen_amd import dist as gdist
    for total in (1, 7, 8, 16, 32, 33):
        for world in (1, 2, 3, 4, 8):
            cuts = [gdist.shard_range(total, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_shards_reproduce_global_batch(tmp_path):
    import json
    from gligen_amd import synthetic as syn
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = {}
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        line = [l for l in out.splitlines() if l.startswith("RESULT ")][-1]
        r = json.loads(line[7:])
        res[r["rank"]] = r
    assert (res[0]["lo"], res[0]["hi"], res[1]["lo"], res[1]["hi"]) == (0, 3, 3, 6)
    assert res[0]["t"] == res[1]["t"] == 2.0          # max over ranks
    assert res[0]["s"] == res[1]["s"] == 6.0          # every sample owned exactly once
    # the two shards are slices of the same seeded global tensors a 1-GPU run of batch 6 would use
    x = syn.make_latent(6, 4, 8, 8, seed=0).double()
    ctx = syn.make_context(6, seed=0).double()
    boxes = syn.make_batch("text", 6, n_valid=8, seed=0)["boxes"].double()
    for r, sl in ((0, slice(0, 3)), (1, slice(3, 6))):
        assert abs(res[r]["x"] - float(x[sl].sum())) < 1e-9
        assert abs(res[r]["ctx"] - float(ctx[sl].sum())) < 1e-9
        assert abs(res[r]["boxes"] - float(boxes[sl].sum())) < 1e-9


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus N` with no launcher around it (the driver's scaling command) must start its N ranks itself:
    re-executed under torch.distributed.run on 127.0.0.1, one JSON line from rank 0 with n_gpus = N and the size of the
    process group the barrier / max-over-ranks clock ran in. --stub swaps the engine for a sleep (no GPU here) and the
    collectives' backend for gloo; the launch path, argument hand-over, barrier and clock are the real ones."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--stub", "--config", "C3"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout          # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["collective_world_size"] == 2 and d["collective_backend"] == "gloo"
    assert d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert "C3" in d["config"]["workload"]     # the arguments reached the ranks
    assert 0 < d["ms_per_step"] < 2000 and d["value"] > 0
    # a launcher / --gpus mismatch is an error, not a silent 1-GPU run
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert bad.returncode != 0 and "must agree" in (bad.stderr + bad.stdout)


# ---- the one exchange of the training path: the trainable set's gradients, gligen_amd.dist.GradBuckets
GRAD_WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, {root!r})
    import torch
    from gligen_amd import dist as gdist
    from ldm.modules.attention import BasicTransformerBlock
    rank, local_rank, world = gdist.init_from_env(backend="gloo")
    blk = BasicTransformerBlock(64, 96, 96, 2, 32, "gatedSA")
    shapes = {{k: tuple(v.shape) for k, v in blk.state_dict().items() if k.startswith("fuser.")}}   # trainer.py:217-245
    gb = gdist.GradBuckets(shapes, bucket_mb=0.05, world=world)    # small buckets: several of them, tensors of very different sizes
    g = torch.Generator().manual_seed(100 + rank)
    for name, v in gb.views.items():
        v.copy_(torch.randn(v.shape, generator=g))
    n_coll = gb.all_reduce(average=True)
    # what every rank must hold now: the mean of the two ranks' seeded gradients
    ref = {{}}
    for r in range(world):
        gr = torch.Generator().manual_seed(100 + r)
        for name, shape in shapes.items():
            t = torch.randn(shape, generator=gr)
            ref[name] = ref.get(name, 0) + t / world
    err = max(float((gb.views[n] - ref[n]).abs().max()) for n in shapes)
    pads = [int(b.numel()) % world for b in gb.buckets]
    split = sum(1 for items in gb.layout for (_, off, n, _) in items if off + n > gb.buckets[gb.layout.index(items)].numel())
    print("RESULT " + json.dumps(dict(rank=rank, err=err, n_buckets=len(gb.buckets), n_coll=n_coll, pads=pads, split=split,
                                      n_tensors=len(shapes), total=sum(int(v.numel()) for v in gb.views.values()))))
    gdist.shutdown()
""")


def test_two_rank_gloo_gradient_buckets(tmp_path):
    import json
    port = _free_port()
    script = tmp_path / "grad_worker.py"
    script.write_text(GRAD_WORKER.format(root=ROOT))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = {}
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        r = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][-1][7:])
        res[r["rank"]] = r
    for r in res.values():
        assert r["err"] < 1e-6                      # both ranks hold the mean gradient of every tensor
        assert r["n_tensors"] == 17 and r["n_buckets"] > 1 and r["n_coll"] == r["n_buckets"]    # one collective per bucket
        assert r["pads"] == [0] * r["n_buckets"] and r["split"] == 0     # bucket lengths divide by the world size, no tensor is split
    assert res[0]["total"] == res[1]["total"]


# ---- the trainer's step over two ranks: gligen_amd.train.TrainStep with a stand-in engine (rank-dependent gradients, a plain AdamW)
TRAIN_WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, {root!r})
    import torch
    from gligen_amd import dist as gdist
    from gligen_amd.train import TrainStep, trainable_names
    rank, local_rank, world = gdist.init_from_env(backend="gloo")

    class Eng:      # what Engine.unet_train_step / op_adamw_step do, on the CPU: every rank sees other data, hence other gradients
        device = torch.device("cpu")
        def unet_train_step(self, cfg, params, batch, fuser_scale=1.0, trainable=None, grads=None, checkpoint=False):
            g = torch.Generator().manual_seed(7 * batch["it"] + rank)
            for k in sorted(grads):
                grads[k].copy_(torch.randn(grads[k].shape, generator=g))
            return torch.tensor([float(rank)]), torch.zeros(1), grads
        waits = []
        def train_wait_grads(self, index, stream=None):      # the overlapped schedule: per bucket wait -> exchange -> update
            self.waits.append(index)
        def op_adamw_step(self, p, g, m, v, step, lr, betas, eps, weight_decay):
            m.mul_(betas[0]).add_(g, alpha=1 - betas[0]); v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
            p.mul_(1 - lr * weight_decay).addcdiv_(m / (1 - betas[0] ** step), (v / (1 - betas[1] ** step)).sqrt() + eps, value=-lr)

    gw = torch.Generator().manual_seed(3)
    sd = {{"input_blocks.1.1.transformer_blocks.0.fuser.linear.weight": torch.randn(16, 8, generator=gw),
          "input_blocks.1.1.transformer_blocks.0.fuser.alpha_attn": torch.randn((), generator=gw),
          "position_net.linears.0.weight": torch.randn(12, 5, generator=gw), "out.2.weight": torch.randn(4, 4, generator=gw)}}
    # rank 1 starts from OTHER trainable values: the constructor broadcasts rank 0's (as DistributedDataParallel does, trainer.py:321-322)
    sd_mine = {{k: (v + 1.0 if (rank == 1 and (".fuser." in k or k.startswith("position_net."))) else v.clone()) for k, v in sd.items()}}
    ts = TrainStep(Eng(), {{}}, sd_mine, lr=0.1, weight_decay=0.01, bucket_mb=4e-4, world=world)
    assert ts.overlap and len(ts.bucket_ready) == len(ts.gbuf.buckets)
    for it in range(3):
        ts.step(dict(it=it))
    assert Eng.waits == ts.bucket_ready * 3      # every bucket waited for its own milestone, in bucket order, every step
    # the same three steps in one process on the MEAN gradient of the two ranks
    names = trainable_names(sd)
    ref = {{k: sd[k].clone() for k in names}}
    opt = torch.optim.AdamW([ref[k].requires_grad_(True) for k in names], lr=0.1, weight_decay=0.01)
    for it in range(3):
        gs = []
        for r in range(world):
            g = torch.Generator().manual_seed(7 * it + r)
            gs.append({{k: torch.randn(ref[k].shape, generator=g) for k in sorted(names)}})
        for k in names:
            ref[k].grad = sum(x[k] for x in gs) / world
        opt.step()
    out = ts.state_dict()
    err = max(float((out[k] - ref[k].detach()).abs().max()) for k in names)
    print("RESULT " + json.dumps(dict(rank=rank, err=err, n_buckets=len(ts.gbuf.buckets), frozen_same=bool(torch.equal(out["out.2.weight"], sd["out.2.weight"])),
                                      digest=float(sum(out[k].double().sum() for k in names)))))
    gdist.shutdown()
""")


def test_two_rank_gloo_train_step(tmp_path):
    """Two ranks with different gradients -- and different initial trainable values, which the constructor's broadcast from rank 0
    levels -- end every step with the same parameters: those of AdamW on the mean gradient
    (the reference's DDP + torch.optim.AdamW, trainer.py:245, 321-322, 384), over several flat buckets."""
    import json
    port = _free_port()
    script = tmp_path / "train_worker.py"
    script.write_text(TRAIN_WORKER.format(root=ROOT))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = {}
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        r = json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][-1][7:])
        res[r["rank"]] = r
    for r in res.values():
        assert r["err"] < 1e-5 and r["frozen_same"] and r["n_buckets"] >= 2, r
    assert res[0]["digest"] == res[1]["digest"]      # bit-identical replicas


def test_gradient_buckets_layout_is_zero_copy():
    """The views handed to the backward kernels ARE the bucket memory (no pack pass), in declaration order, one bucket for the
    whole set when it fits."""
    from gligen_amd.dist import GradBuckets
    shapes = {"a.weight": (8, 4), "a.bias": (8,), "b.weight": (16, 8)}
    gb = GradBuckets(shapes, bucket_mb=1.0, world=8)
    assert len(gb.buckets) == 1 and gb.buckets[0].numel() % 8 == 0 and gb.buckets[0].numel() >= 32 + 8 + 128
    gb.views["a.bias"].fill_(3.0)
    assert float(gb.buckets[0][32:40].sum()) == 24.0 and float(gb.buckets[0].sum()) == 24.0
    assert [n for items in gb.layout for (n, *_rest) in items] == list(shapes)
    assert GradBuckets(shapes, bucket_mb=1e-4, world=2).all_reduce() == 0      # no process group: nothing to exchange
