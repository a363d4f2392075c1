"""Pin the CPU oracle (oracle/gligen_oracle.py) to outputs of the REAL reference stored under
tests/golden/ by oracle/make_golden.py. No GPU. fp32 vs fp32: only op-ordering noise is allowed."""
import os

import numpy as np
import pytest
import torch

from helpers import golden_shapes, grounding_kwargs, load_golden, mse, oracle_cfg, unet_inputs
from gligen_amd import synthetic as syn
from oracle import gligen_oracle as orc

FP32_TOL = 1e-9  # MSE between two fp32 evaluations of the same graph


@pytest.mark.parametrize("name", ["unet_small_text", "unet_small_text_image", "unet_small_keypoint", "unet_small_inpaint", "unet_small_gatedca",
                                  "unet_small_gatedsa2"])
def test_unet_small(name):
    g = load_golden(name)
    meta = g["meta"]
    sd = syn.seeded_state_dict(golden_shapes(name), meta["weight_seed"])
    batch, x, ctx, t, extra = unet_inputs(meta)
    gk = grounding_kwargs(meta["kind"], batch)
    cfg = oracle_cfg(meta["cfg"], meta["kind"])
    inp = dict(x=x, timesteps=t, context=ctx, grounding_input=gk, inpainting_extra_input=extra)
    with torch.no_grad():
        assert mse(orc.position_net(sd, "position_net", meta["kind"], gk), g["objs"]) < FP32_TOL
        assert mse(orc.unet_forward(sd, cfg, inp), g["eps"]) < FP32_TOL
        inp_null = dict(inp, grounding_input=orc.null_grounding(meta["kind"], gk))
        assert mse(orc.unet_forward(sd, cfg, inp_null), g["eps_null"]) < FP32_TOL
        # the reference's set_alpha_scale matches GatedSelfAttentionDense / GatedCrossAttentionDense by exact type
        # (gligen_inference.py:24-28) and therefore never reaches a GatedSelfAttentionDense2: its scale stays 1
        sa2 = cfg["fuser_type"] == "gatedSA2"
        assert mse(orc.unet_forward(sd, cfg, inp, fuser_scale=1.0 if sa2 else 0.3), g["eps_scale03"]) < FP32_TOL
    # the fixture is not vacuous: grounding and the gate scale change eps measurably
    if sa2:
        assert np.array_equal(g["eps"], g["eps_scale03"]) and mse(g["eps"], g["eps_null"]) > 1e-4
    else:
        assert mse(g["eps"], g["eps_scale03"]) > 1e-4
    assert g["eps"].std() > 0.1


def test_unet_full():
    """All 966 tensors / 1.07 B parameters of the shipped SD-1.4 GLIGEN UNet, latent 16x16."""
    g = load_golden("unet_full_text")
    meta = g["meta"]
    shapes = golden_shapes("unet_full_text")
    assert len(shapes) == 966 and meta["n_params"] == 1068623204
    sd = syn.seeded_state_dict(shapes, meta["weight_seed"])
    batch, x, ctx, t, extra = unet_inputs(meta)
    inp = dict(x=x, timesteps=t, context=ctx, grounding_input=grounding_kwargs("text", batch))
    with torch.no_grad():
        assert mse(orc.unet_forward(sd, oracle_cfg(meta["cfg"], "text"), inp), g["eps"]) < FP32_TOL


@pytest.mark.parametrize("name", ["vae_small", "vae_full"])
def test_vae_decode(name):
    g = load_golden(name)
    meta = g["meta"]
    sd = syn.seeded_state_dict(golden_shapes(name), meta["weight_seed"])
    z = syn.make_latent(meta["B"], 4, meta["hw"], meta["hw"], seed=3) * 0.18215 * 4
    dd = meta["ddconfig"]
    with torch.no_grad():
        img = orc.vae_decode(sd, dict(ch_mult=dd["ch_mult"], num_res_blocks=dd["num_res_blocks"], scale_factor=0.18215), z)
    assert mse(img, g["img"]) < 1e-8


def test_schedule_and_small_functions():
    m = load_golden("misc")
    sched = orc.make_schedule()
    np.testing.assert_allclose(sched["alphas_cumprod"], m["diff_alphas_cumprod"], rtol=1e-6)
    np.testing.assert_allclose(sched["sqrt_alphas_cumprod"], m["diff_sqrt_alphas_cumprod"], rtol=1e-6)
    np.testing.assert_allclose(sched["sqrt_one_minus_alphas_cumprod"], m["diff_sqrt_one_minus_alphas_cumprod"], rtol=1e-6)
    for S in (20, 50):
        for tp, tag in ((None, "none"), ([0.3, 0.0, 0.7], "0.3_0.0_0.7"), ([0.5, 0.25, 0.25], "0.5_0.25_0.25")):
            np.testing.assert_allclose(np.asarray(orc.alpha_generator(S, tp), dtype=np.float64), m[f"alpha_{S}_{tag}"])
    boxes = torch.from_numpy(m["mask_boxes"])
    assert np.array_equal(orc.draw_masks_from_boxes(boxes, 64).numpy(), m["mask64"])
    np.testing.assert_allclose(orc.timestep_embedding(torch.tensor([1, 441, 981]), 320).numpy(), m["temb"], atol=1e-6)
    for name in ("plms_trace_50", "plms_trace_20"):
        tr = load_golden(name)
        ps = orc.plms_schedule(tr["meta"]["S"], sched)
        assert np.array_equal(ps["ddim_timesteps"], tr["ddim_timesteps"])
        np.testing.assert_allclose(ps["ddim_alphas"], tr["ddim_alphas"], rtol=1e-6)
        np.testing.assert_allclose(ps["ddim_alphas_prev"], tr["ddim_alphas_prev"], rtol=1e-6)


@pytest.mark.parametrize("name", ["plms_trace_50", "plms_trace_20"])
def test_plms_loop_against_reference_trace(name):
    """Same call sequence (timestep, cond/uncond, gate scale) and same final latent as the reference sampler
    driving a cheap mock model (102 calls for S=50, 42 for S=20)."""
    tr = load_golden(name)
    S, atype = tr["meta"]["S"], tr["meta"]["alpha_type"]
    calls = []

    def eps_fn(x, t, cond, scale):
        calls.append((int(t[0]), bool(cond), float(scale)))
        return torch.tanh(x) * (0.5 if cond else 0.3) + 0.01 * t.float().view(-1, 1, 1, 1) / 1000

    x = syn.make_latent(2, 4, 8, 8, seed=5)
    out = orc.plms_sample(eps_fn, x, S, orc.make_schedule(), 7.5, alphas=orc.alpha_generator(S, atype))
    ref_calls = [(int(a), bool(b), float(c)) for a, b, c in tr["calls"]]
    assert calls == ref_calls
    assert mse(out, tr["x_out"]) < 1e-10


@pytest.mark.parametrize("name", ["plms_unet_small", "plms_unet_small_inpaint"])
def test_plms_with_unet(name):
    g = load_golden(name)
    meta = g["meta"]
    shapes = golden_shapes("unet_small_inpaint" if meta["inpaint"] else "unet_small_text")
    sd = syn.seeded_state_dict(shapes, 1234)
    B, hw, S = meta["B"], meta["hw"], meta["S"]
    batch = syn.make_batch("text", B, n_valid=meta["n_valid"], seed=1)
    gk = grounding_kwargs("text", batch)
    gnull = orc.null_grounding("text", gk)
    ctx, uc = syn.make_context(B, seed=1), syn.make_context(B, seed=9)
    cfg = oracle_cfg(syn.UNET_CFG_SMALL, "text")
    mask = z0 = extra = noise = None
    if meta["inpaint"]:
        mask = orc.draw_masks_from_boxes(batch["boxes"], hw)
        z0 = syn.make_latent(B, 4, hw, hw, seed=2)
        extra = torch.cat([z0 * mask, mask], dim=1)
        noise = torch.from_numpy(g["noise"])

    def eps_fn(x, t, cond, scale):
        inp = dict(x=x, timesteps=t, context=ctx if cond else uc, grounding_input=gk if cond else gnull, inpainting_extra_input=extra)
        return orc.unet_forward(sd, cfg, inp, fuser_scale=scale)

    def swap_in_sd_first_conv():  # model.restore_first_conv_from_SD(); skipped for inpainting models
        if not meta["inpaint"]:
            sdc = syn.sd_first_conv_state()
            sd["input_blocks.0.0.weight"], sd["input_blocks.0.0.bias"] = sdc["weight"], sdc["bias"]

    with torch.no_grad():
        out = orc.plms_sample(eps_fn, syn.make_latent(B, 4, hw, hw, seed=6), S, orc.make_schedule(), 7.5,
                              alphas=orc.alpha_generator(S, meta["alpha_type"]), mask=mask, x0=z0, noise=noise,
                              on_gate_off=swap_in_sd_first_conv)
    assert mse(out, g["x_out"]) < 1e-7


def test_ddim_loop_against_reference_trace():
    """DDIMSampler (eta 0) of the reference driving the cheap mock model: same call sequence, same final latent."""
    tr = load_golden("ddim_trace_25")
    S, atype = tr["meta"]["S"], tr["meta"]["alpha_type"]
    calls = []

    def eps_fn(x, t, cond, scale):
        calls.append((int(t[0]), bool(cond), float(scale)))
        return torch.tanh(x) * (0.5 if cond else 0.3) + 0.01 * t.float().view(-1, 1, 1, 1) / 1000

    x = syn.make_latent(2, 4, 8, 8, seed=5)
    out = orc.ddim_sample(eps_fn, x, S, orc.make_schedule(), 7.5, alphas=orc.alpha_generator(S, atype))
    assert calls == [(int(a), bool(b), float(c)) for a, b, c in tr["calls"]]
    assert len(calls) == 2 * S
    assert mse(out, tr["x_out"]) < 1e-10


@pytest.mark.parametrize("name", ["ddim_unet_small", "ddim_unet_small_inpaint"])
def test_ddim_with_unet(name):
    g = load_golden(name)
    meta = g["meta"]
    shapes = golden_shapes("unet_small_inpaint" if meta["inpaint"] else "unet_small_text")
    sd = syn.seeded_state_dict(shapes, 1234)
    B, hw, S = meta["B"], meta["hw"], meta["S"]
    batch = syn.make_batch("text", B, n_valid=meta["n_valid"], seed=1)
    gk = grounding_kwargs("text", batch)
    gnull = orc.null_grounding("text", gk)
    ctx, uc = syn.make_context(B, seed=1), syn.make_context(B, seed=9)
    cfg = oracle_cfg(syn.UNET_CFG_SMALL, "text")
    mask = z0 = extra = noise = None
    if meta["inpaint"]:
        mask = orc.draw_masks_from_boxes(batch["boxes"], hw)
        z0 = syn.make_latent(B, 4, hw, hw, seed=2)
        extra = torch.cat([z0 * mask, mask], dim=1)
        noise = torch.from_numpy(g["noise"])

    def eps_fn(x, t, cond, scale):
        inp = dict(x=x, timesteps=t, context=ctx if cond else uc, grounding_input=gk if cond else gnull, inpainting_extra_input=extra)
        return orc.unet_forward(sd, cfg, inp, fuser_scale=scale)

    def swap_in_sd_first_conv():
        if not meta["inpaint"]:
            sdc = syn.sd_first_conv_state()
            sd["input_blocks.0.0.weight"], sd["input_blocks.0.0.bias"] = sdc["weight"], sdc["bias"]

    sched = orc.make_schedule()
    n_steps = len(orc.plms_schedule(S, sched)["ddim_timesteps"])  # S = 6 -> arange(0, 1000, 166) has 7 entries (util.py:55-69)
    with torch.no_grad():
        out = orc.ddim_sample(eps_fn, syn.make_latent(B, 4, hw, hw, seed=6), S, sched, 7.5,
                              alphas=orc.alpha_generator(n_steps, meta["alpha_type"]), mask=mask, x0=z0, noise=noise,
                              on_gate_off=swap_in_sd_first_conv)
    assert mse(out, g["x_out"]) / float(g["x_out"].var()) < 1e-8


@pytest.mark.parametrize("name", ["vae_enc_small", "vae_enc_full"])
def test_vae_encode(name):
    g = load_golden(name)
    dd = g["meta"]["ddconfig"]
    sd = syn.seeded_state_dict(golden_shapes("vae_small" if name.endswith("small") else "vae_full"), 4321)
    x = torch.rand(g["meta"]["B"], 3, g["meta"]["res"], g["meta"]["res"], generator=torch.Generator().manual_seed(8)) * 2 - 1
    with torch.no_grad():
        z = orc.vae_encode(sd, dict(ch_mult=dd["ch_mult"], num_res_blocks=dd["num_res_blocks"], scale_factor=0.18215), x,
                           torch.from_numpy(g["noise"]))
    assert mse(z, g["z"]) / float(g["z"].var()) < 1e-9


def test_bicubic_taps_of_the_resize_kernel():
    """The arithmetic of misc.hip:fuser_resize_kernel (source index, cubic-convolution taps with A = -0.75, clamped taps),
    restated in numpy, against torch's F.interpolate(mode='bicubic') which the reference calls (attention.py:291)."""
    import torch.nn.functional as F

    def taps(t, A=-0.75):
        a, b, c = t + 1.0, 1.0 - t, 2.0 - t
        return [((A * a - 5 * A) * a + 8 * A) * a - 4 * A, ((A + 2) * t - (A + 3)) * t * t + 1,
                ((A + 2) * b - (A + 3)) * b * b + 1, ((A * c - 5 * A) * c + 8 * A) * c - 4 * A]

    def resize(inp, so):
        si = inp.shape[-1]
        out = np.zeros(inp.shape[:2] + (so, so), np.float32)
        sc = si / so
        for y in range(so):
            fy = sc * (y + 0.5) - 0.5
            iy = int(np.floor(fy))
            wy = taps(fy - iy)
            for x in range(so):
                fx = sc * (x + 0.5) - 0.5
                ix = int(np.floor(fx))
                wx = taps(fx - ix)
                for p in range(4):
                    yy = min(max(iy - 1 + p, 0), si - 1)
                    for q in range(4):
                        xx = min(max(ix - 1 + q, 0), si - 1)
                        out[:, :, y, x] += wy[p] * wx[q] * inp[:, :, yy, xx]
        return out

    x = torch.randn(2, 3, 4, 4, generator=torch.Generator().manual_seed(0))
    for so in (2, 4, 8, 16, 64):
        assert np.abs(F.interpolate(x, (so, so), mode="bicubic").numpy() - resize(x.numpy(), so)).max() < 2e-6


DOWNSAMPLER = {"canny": dict(n_in=1, mode="bicubic"), "hed": dict(n_in=1, mode="bicubic"), "normal": dict(n_in=3, mode="bicubic"),
               "sem": dict(n_in=152, mode="nearest"), "depth": dict(n_in=1, mode="bicubic")}


@pytest.mark.parametrize("modality", ["canny", "hed", "normal", "sem", "depth"])
def test_spatial_modalities(modality):
    """ConvNeXt tokenizer, GroundingDownsampler and the 4 + k channel first conv of the spatial-map modalities against the
    reference's outputs (oracle/make_golden.py:spatial_case)."""
    name = f"unet_small_{modality}"
    g = load_golden(name)
    meta = g["meta"]
    sd = syn.seeded_state_dict(golden_shapes(name), meta["weight_seed"])
    B, hw = meta["B"], meta["hw"]
    img = syn.make_spatial_map(modality, B, meta["res"], seed=1)
    tk, dsp = meta["cfg"]["grounding_tokenizer"]["params"], meta["cfg"]["grounding_downsampler"]["params"]
    ds_cfg = dict(DOWNSAMPLER[modality], resize=dsp.get("resize_input", 64))
    cfg = dict(oracle_cfg(meta["cfg"], "spatial"), tok_resize=tk["resize_input"], downsampler=ds_cfg)
    x, ctx = syn.make_latent(B, 4, hw, hw, seed=1), syn.make_context(B, seed=1)
    t = torch.tensor([981, 441][:B], dtype=torch.long)
    with torch.no_grad():
        objs = orc.spatial_position_net(sd, "position_net", img, torch.ones(B, 1), tk["resize_input"])
        assert mse(objs, g["objs"].astype(np.float32)) < 1e-6          # golden tokens are stored as float16
        objs_null = orc.spatial_position_net(sd, "position_net", torch.zeros_like(img), torch.zeros(B), tk["resize_input"])
        assert mse(objs_null, g["objs_null"].astype(np.float32)) < 1e-6
        ds = orc.grounding_downsampler(sd, "downsample_net", img, ds_cfg["n_in"], ds_cfg["resize"], ds_cfg["mode"])
        assert mse(ds, g["ds"]) < FP32_TOL
        inp = dict(x=x, timesteps=t, context=ctx, grounding_input=dict(image=img, mask=torch.ones(B, 1)), grounding_extra_input=img)
        assert mse(orc.unet_forward(sd, cfg, inp), g["eps"]) < FP32_TOL
        inp_null = dict(inp, grounding_input=dict(image=torch.zeros_like(img), mask=torch.zeros(B)))
        assert mse(orc.unet_forward(sd, cfg, inp_null), g["eps_null"]) < FP32_TOL
    assert mse(g["eps"], g["eps_null"]) > 1e-5


@pytest.mark.parametrize("name", ["resblock_backward_skipconv", "resblock_backward_identity"])
def test_resblock_backward_golden(name):
    """The ResBlock training-slice goldens (the reference's autograd): the oracle's ResBlock reproduces the forward, and autograd
    through the oracle reproduces dL/dx -- which pins the oracle as a checker for the backward as well."""
    import json
    from helpers import GOLDEN, golden_shapes, resblock_backward_inputs
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(g["meta"]).decode())
    x, emb, target = resblock_backward_inputs(meta)
    assert abs(float(x.double().sum()) - float(g["x_sum"])) < 1e-6 and abs(float(target.double().sum()) - float(g["target_sum"])) < 1e-6
    sd = {"rb." + k: v for k, v in syn.seeded_state_dict({k: tuple(v) for k, v in golden_shapes(name).items()}, meta["seed"]).items()}
    x = x.requires_grad_(True)
    y = orc.unet_resblock(sd, "rb", x, emb)
    loss = torch.nn.functional.mse_loss(y, target)
    loss.backward()
    assert float((y.detach() - torch.from_numpy(g["y"])).abs().max()) < 1e-4
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    ref = torch.from_numpy(g["dx"])
    assert float(((x.grad - ref) ** 2).mean() / (ref ** 2).mean()) < 1e-8


@pytest.mark.parametrize("case", ["unet_small_train_step", "unet_small_ti_train_step", "unet_small_kp_train_step"])
def test_unet_train_step_golden_through_the_oracle(case):
    """The whole-iteration goldens (the reference's loss.backward() on the small UNet, for the text, text+image and keypoint
    tokenizers): torch autograd THROUGH the oracle's unet_forward reproduces the loss and the sampled gradients of every trainable
    tensor -- the oracle is a pinned checker for the training path too (smoke() uses it that way)."""
    g = load_golden(case)
    meta = g["meta"]
    B, hw, kind = meta["B"], meta["hw"], meta.get("kind", "text")
    sd = syn.seeded_state_dict(golden_shapes(case), meta["weight_seed"])
    train = [k for k in sd if ".fuser." in k or k.startswith("position_net.")]
    assert len(train) == meta["n_trainable"]
    for k in train:
        sd[k].requires_grad_(True)
    b = syn.make_batch(kind, B, n_valid=meta["n_valid"], seed=5)
    if kind == "text_image":
        b["text_masks"][:, 1] = 0
        b["image_masks"][:, 0] = 0
    inp = dict(x=syn.make_latent(B, 4, hw, hw, seed=6), timesteps=torch.tensor([981, 441][:B]), context=syn.make_context(B, seed=6),
               grounding_input=grounding_kwargs(kind, b))
    eps = orc.unet_forward(sd, oracle_cfg(meta["cfg"], kind), inp)
    loss = torch.nn.functional.mse_loss(eps, syn.make_latent(B, 4, hw, hw, seed=7))
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    n = meta["sample"]
    worst = 0.0
    for k in train:
        flat = sd[k].grad.reshape(-1)
        stride = max(1, flat.numel() // n)
        sub = flat[::stride][:n] if flat.numel() > n else flat
        ref = torch.from_numpy(g["grad." + k].astype(np.float32)) * float(g["scale." + k])
        if ref.numel() == 1:
            assert abs(float(sub) - float(ref)) < 2e-3 * max(abs(float(ref)), 1e-4), k
            continue
        worst = max(worst, float(((sub - ref) ** 2).mean() / (ref ** 2).mean()))
    assert worst < 1e-5, worst        # (the golden stores fp16 samples: 1e-7)
