"""End-to-end parity of the HIP path (drop-in ldm.* modules -> C ABI -> kernels) on a real MI355X.

Checked against (a) the committed outputs of the REAL reference (tests/golden, made by
oracle/make_golden.py) and (b) the CPU oracle at sizes it finishes in seconds. The engine computes
in bf16 storage / fp32 accumulate, the reference in fp32, so tolerances are stated as eps-MSE:
north-star bar 1e-3; torch's own bf16-autocast error on this kind of fixture is ~8e-5.
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import (ROOT, build_product_unet, build_product_vae, golden_shapes, grounding_kwargs, load_golden, mse,
                     oracle_cfg, scripted_randn_like, unet_inputs)
from gligen_amd import synthetic as syn

pytestmark = pytest.mark.gpu

EPS_MSE_TOL = 2e-4     # absolute, on eps with std ~0.28  (bar in BASELINE.json: 1e-3)
IMG_MSE_TOL = 2e-3     # decoded image in [-2.9, 2.9] (std ~0.6) through ~30 bf16 layers
REPORT = {}


@pytest.fixture(scope="module", autouse=True)
def _report():
    yield
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _to(d, dev):
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in d.items()}


@pytest.mark.parametrize("name", ["unet_small_text", "unet_small_text_image", "unet_small_keypoint", "unet_small_inpaint", "unet_small_gatedca",
                                  "unet_small_gatedsa2"])
def test_unet_small_vs_reference(name):
    dev = _dev()
    g = load_golden(name)
    meta = g["meta"]
    model = build_product_unet(meta["cfg"], meta["kind"], meta["inpaint"], device=dev)
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == golden_shapes(name)
    batch, x, ctx, t, extra = unet_inputs(meta)
    gin = model.grounding_tokenizer_input.prepare(_to(batch, dev))
    inp = dict(x=x.to(dev), timesteps=t.to(dev), context=ctx.to(dev), grounding_input=gin,
               inpainting_extra_input=None if extra is None else extra.to(dev), grounding_extra_input=None)
    from gligen_inference import set_alpha_scale
    eps = model(inp)
    eps_null = model({k: v for k, v in inp.items() if k != "grounding_input"})
    set_alpha_scale(model, 0.3)
    eps_s = model(inp)
    set_alpha_scale(model, 1)
    r = dict(eps=mse(eps, g["eps"]), eps_null=mse(eps_null, g["eps_null"]), eps_scale03=mse(eps_s, g["eps_scale03"]),
             eps_var=float(g["eps"].var()))
    # the grounding / gate effects themselves must be reproduced, not just the bulk of eps
    if meta["cfg"].get("fuser_type") == "gatedSA2":
        # reference quirk: set_alpha_scale never reaches GatedSelfAttentionDense2 (exact-type match, gligen_inference.py:24-28);
        # the grounding effect is checked instead
        assert torch.equal(eps_s, eps) and np.array_equal(g["eps"], g["eps_scale03"])
        d_ref = torch.from_numpy(g["eps"] - g["eps_null"])
        d_hip = (eps - eps_null).float().cpu()
    else:
        d_ref = torch.from_numpy(g["eps"] - g["eps_scale03"])
        d_hip = (eps - eps_s).float().cpu()
    r["gate_effect_rel_err"] = float(((d_hip - d_ref) ** 2).mean() / (d_ref ** 2).mean())
    REPORT[name] = r
    assert eps.shape == tuple(g["eps"].shape) and eps.dtype == torch.float32 and eps.device.type == "cuda"
    assert r["eps"] < EPS_MSE_TOL and r["eps_null"] < EPS_MSE_TOL and r["eps_scale03"] < EPS_MSE_TOL, r
    # (gatedSA2: the effect checked is eps - eps_null, an order of magnitude smaller than the gate-scale effect of the
    #  other fixtures, so bf16 noise weighs more against it)
    assert r["gate_effect_rel_err"] < (0.15 if meta["cfg"].get("fuser_type") == "gatedSA2" else 0.05), r
    # determinism: same inputs -> bit-identical eps
    assert torch.equal(model(inp), eps)


def test_per_fuser_scales():
    """The fusers' `scale` attributes are plain per-module values in the reference (attention.py:231); set individually they
    reach the engine one by one (gl_unet_set_fuser_scales), a zero among them included."""
    dev = _dev()
    from ldm.modules.attention import GatedSelfAttentionDense
    from oracle import gligen_oracle as orc
    g = load_golden("unet_small_text")
    meta = g["meta"]
    model = build_product_unet(meta["cfg"], "text", device=dev)
    batch, x, ctx, t, _ = unet_inputs(meta)
    fusers = [m for m in model.modules() if type(m) is GatedSelfAttentionDense]
    scales = [0.3, 1.0, 0.0, 0.7, 1.5][:len(fusers)] + [1.0] * max(0, len(fusers) - 5)
    for m, sc in zip(fusers, scales):
        m.scale = sc
    assert model.fuser_scales() == scales
    with pytest.raises(ValueError):
        model.fuser_scale()
    gin = model.grounding_tokenizer_input.prepare(_to(batch, dev))
    eps = model(dict(x=x.to(dev), timesteps=t.to(dev), context=ctx.to(dev), grounding_input=gin, inpainting_extra_input=None,
                     grounding_extra_input=None))
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref = orc.unet_forward(sd, oracle_cfg(meta["cfg"], "text"), dict(x=x, timesteps=t, context=ctx, grounding_input=grounding_kwargs("text", batch)),
                               fuser_scale=scales)
        ref_uniform = orc.unet_forward(sd, oracle_cfg(meta["cfg"], "text"), dict(x=x, timesteps=t, context=ctx,
                                                                                 grounding_input=grounding_kwargs("text", batch)))
    REPORT["per_fuser_scales"] = dict(eps=mse(eps, ref), vs_uniform=mse(ref, ref_uniform))
    assert mse(ref, ref_uniform) > 1e-4          # the individual scales matter
    assert mse(eps, ref) < EPS_MSE_TOL, REPORT["per_fuser_scales"]
    # all fusers at 0: the fuser branches are skipped outright, which is exact
    for m in fusers:
        m.scale = 0
    eps0 = model(dict(x=x.to(dev), timesteps=t.to(dev), context=ctx.to(dev), grounding_input=gin, inpainting_extra_input=None,
                      grounding_extra_input=None))
    with torch.no_grad():
        ref0 = orc.unet_forward(sd, oracle_cfg(meta["cfg"], "text"), dict(x=x, timesteps=t, context=ctx, grounding_input=grounding_kwargs("text", batch)),
                                fuser_scale=0.0)
    assert mse(eps0, ref0) < EPS_MSE_TOL


def test_unet_full_vs_reference():
    """The shipped topology (966 tensors, 1.07 B params) at latent 16x16 against the reference's output."""
    dev = _dev()
    g = load_golden("unet_full_text")
    meta = g["meta"]
    model = build_product_unet(meta["cfg"], "text", device=dev)
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == golden_shapes("unet_full_text")
    batch, x, ctx, t, _ = unet_inputs(meta)
    gin = model.grounding_tokenizer_input.prepare(_to(batch, dev))
    eps = model(dict(x=x.to(dev), timesteps=t.to(dev), context=ctx.to(dev), grounding_input=gin,
                     inpainting_extra_input=None, grounding_extra_input=None))
    REPORT["unet_full_text_16"] = dict(eps=mse(eps, g["eps"]), eps_var=float(g["eps"].var()))
    assert mse(eps, g["eps"]) < EPS_MSE_TOL, REPORT["unet_full_text_16"]

    # full 512x512 size (latent 64x64, B=1, Ng=30) against the CPU oracle
    from oracle import gligen_oracle as orc
    B, hw = 1, 64
    batch = syn.make_batch("text", B, n_valid=8, seed=3)
    x, ctx, t = syn.make_latent(B, 4, hw, hw, seed=3), syn.make_context(B, seed=3), torch.tensor([501])
    gk = grounding_kwargs("text", batch)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref = orc.unet_forward(sd, oracle_cfg(meta["cfg"], "text"), dict(x=x, timesteps=t, context=ctx, grounding_input=gk))
    gin = model.grounding_tokenizer_input.prepare(_to(batch, dev))
    eps = model(dict(x=x.to(dev), timesteps=t.to(dev), context=ctx.to(dev), grounding_input=gin,
                     inpainting_extra_input=None, grounding_extra_input=None))
    REPORT["unet_full_text_64"] = dict(eps=mse(eps, ref), eps_var=float(ref.var()))
    assert mse(eps, ref) < EPS_MSE_TOL, REPORT["unet_full_text_64"]


def test_unet_full_large_batches_agree_with_the_benchmark_batch():
    """run() keeps a per-GPU batch whole below 32 images (round 6), so evaluations of 16 and 32 samples at 64 x 64 -- other tiles, K splits
    and row-local policies than the benchmark's 8 -- are a product path: sample i of the large evaluation must be sample i of the
    8-sample evaluation that holds it, to the bf16 noise of two different tilings (either path sits ~1.7e-4 relative from the fp32
    reference: test_unet_full_vs_reference; measured between the two: 1.5e-4)."""
    dev = _dev()
    meta = load_golden("unet_full_text")["meta"]
    model = build_product_unet(meta["cfg"], "text", device=dev)
    hw = 64
    for n in (16, 32):
        batch = syn.make_batch("text", n, n_valid=8, seed=5)
        x, ctx = syn.make_latent(n, 4, hw, hw, seed=5), syn.make_context(n, seed=5)
        t = torch.tensor([981, 741, 501, 261] * (n // 4))

        def run(lo, hi):
            sub = {k: (v[lo:hi] if torch.is_tensor(v) and v.shape[0] == n else v) for k, v in batch.items()}
            gin = model.grounding_tokenizer_input.prepare(_to(sub, dev))
            return model(dict(x=x[lo:hi].to(dev), timesteps=t[lo:hi].to(dev), context=ctx[lo:hi].to(dev), grounding_input=gin,
                              inpainting_extra_input=None, grounding_extra_input=None)).float().cpu()
        whole = run(0, n)
        parts = torch.cat([run(i, i + 8) for i in range(0, n, 8)])
        rel = mse(whole, parts) / float(parts.var())
        REPORT[f"unet_full_text_64_batch{n}_vs_8"] = dict(rel_mse=rel)
        assert bool(torch.isfinite(whole).all()) and rel < 5e-4, REPORT[f"unet_full_text_64_batch{n}_vs_8"]
    model._drop_engine()


@pytest.mark.parametrize("name,dd", [("vae_small", "VAE_DDCONFIG_SMALL"), ("vae_full", "VAE_DDCONFIG")])
def test_vae_decode_vs_reference(name, dd):
    dev = _dev()
    g = load_golden(name)
    meta = g["meta"]
    ae = build_product_vae(getattr(syn, dd), device=dev)
    assert {k: list(v.shape) for k, v in ae.state_dict().items()} == golden_shapes(name)
    z = syn.make_latent(meta["B"], 4, meta["hw"], meta["hw"], seed=3) * 0.18215 * 4
    img = ae.decode(z.to(dev))
    REPORT[name] = dict(img=mse(img, g["img"]), img_var=float(g["img"].var()))
    assert img.shape == tuple(g["img"].shape)
    assert mse(img, g["img"]) < IMG_MSE_TOL, REPORT[name]
    if name == "vae_full":  # 512x512 decode against the oracle + uint8 epilogue
        from oracle import gligen_oracle as orc
        z = syn.make_latent(1, 4, 64, 64, seed=4) * 0.18215 * 4
        sd = {k: v.detach().cpu() for k, v in ae.state_dict().items()}
        d = getattr(syn, dd)
        with torch.no_grad():
            ref = orc.vae_decode(sd, dict(ch_mult=d["ch_mult"], num_res_blocks=d["num_res_blocks"], scale_factor=0.18215), z)
        img = ae.decode(z.to(dev))
        REPORT["vae_full_512"] = dict(img=mse(img, ref), img_var=float(ref.var()))
        assert mse(img, ref) < IMG_MSE_TOL, REPORT["vae_full_512"]
        u8 = ae.engine.to_uint8(img).cpu().numpy()
        assert np.array_equal(u8, orc.to_uint8(img.cpu())), "uint8 epilogue must be bit-exact on the same float image"
        diff = np.abs(u8.astype(np.int32) - orc.to_uint8(ref).astype(np.int32))
        REPORT["vae_full_512"]["u8_mean_abs_diff"] = float(diff.mean())
        assert diff.mean() < 3.0


def test_vae_decode_large_batch_goes_through_in_slices():
    """A batch beyond the arena's sizing (8 images at 64 x 64 latents) is decoded in slices of the batch: image i of a batch of 10 is
    image i of its slice bit for bit, and the tail slice (2 images: other tile choices) agrees within the decode tolerance with the
    same latents decoded among 8."""
    dev = _dev()
    ae = build_product_vae(syn.VAE_DDCONFIG, device=dev)
    z = (syn.make_latent(10, 4, 64, 64, seed=21) * 0.18215 * 4).to(dev)
    img = ae.decode(z)
    assert img.shape == (10, 3, 512, 512) and bool(torch.isfinite(img).all())
    assert torch.equal(img[:8], ae.decode(z[:8])) and torch.equal(img[8:], ae.decode(z[8:]))
    among8 = ae.decode(torch.cat([z[8:], z[:6]]))[:2]
    assert mse(img[8:], among8) < 1e-4 * float(among8.var()), mse(img[8:], among8)
    ae._drop_engine()


@pytest.mark.parametrize("name,dd", [("vae_enc_small", "VAE_DDCONFIG_SMALL"), ("vae_enc_full", "VAE_DDCONFIG")])
def test_vae_encode_vs_reference(name, dd, monkeypatch):
    """AutoencoderKL.encode (inpainting: once per prompt) on the device vs the reference's output for the same image, the
    same seeded weights and the same posterior noise draw."""
    dev = _dev()
    g = load_golden(name)
    ae = build_product_vae(getattr(syn, dd), device=dev)
    x = torch.rand(g["meta"]["B"], 3, g["meta"]["res"], g["meta"]["res"], generator=torch.Generator().manual_seed(8)) * 2 - 1
    noise = torch.from_numpy(g["noise"])
    monkeypatch.setattr(torch, "randn", lambda *a, **k: noise.clone())  # the one posterior draw, as recorded
    z = ae.encode(x.to(dev))
    monkeypatch.undo()
    rel = mse(z, g["z"]) / float(g["z"].var())
    REPORT[name] = dict(z_rel_mse=rel)
    assert z.shape == tuple(g["z"].shape) and rel < 5e-3, REPORT[name]
    ae._drop_engine()


PLMS_TOL = 2e-3   # relative MSE of the final latent (measured on MI355X: 3e-5 .. 2.5e-4 over the seven cases)


@pytest.mark.parametrize("name", ["plms_unet_small", "plms_unet_small_inpaint", "ddim_unet_small", "ddim_unet_small_inpaint",
                                  "plms_unet_small_gatedsa2", "plms_unet_small_inpaint_x0b1", "plms50_unet_small"])
def test_plms_vs_reference(name, tmp_path, monkeypatch):
    """Sampler-level parity with the reference's own PLMS / DDIM loops: alpha schedule + SD first-conv swap, inpainting
    blend (incl. one encoded image broadcast over a larger batch, as run() does), the gatedSA2 fuser under a schedule
    (the reference never rescales that class), and a full 50-step run (51 graph replays)."""
    dev = _dev()
    from functools import partial
    from gligen_inference import alpha_generator, set_alpha_scale
    from ldm.models.diffusion.ldm import LatentDiffusion
    from ldm.models.diffusion.plms import PLMSSampler
    if name.startswith("ddim"):  # the reference's DDIMSampler (eta 0) behind the same device loop
        from ldm.models.diffusion.ddim import DDIMSampler as PLMSSampler  # noqa: F811
    from oracle.gligen_oracle import draw_masks_from_boxes
    g = load_golden(name)
    meta = g["meta"]
    B, hw, S = meta["B"], meta["hw"], meta["S"]
    torch.save(syn.sd_first_conv_state(), tmp_path / "SD_input_conv_weight_bias.pth")
    monkeypatch.chdir(tmp_path)
    diffusion = LatentDiffusion(linear_start=0.00085, linear_end=0.012, timesteps=1000).to(dev)
    cfg = meta.get("cfg", syn.UNET_CFG_SMALL)
    batch = syn.make_batch("text", B, n_valid=meta["n_valid"], seed=1, max_objs=meta.get("max_objs", 30))
    ctx, uc = syn.make_context(B, seed=1).to(dev), syn.make_context(B, seed=9).to(dev)
    results = []
    for use_graph in (False, True):
        model = build_product_unet(cfg, "text", meta["inpaint"], device=dev)
        gin = model.grounding_tokenizer_input.prepare(_to(batch, dev))
        mask = z0 = extra = None
        sampler = PLMSSampler(diffusion, model, alpha_generator_func=partial(alpha_generator, type=meta["alpha_type"]),
                              set_alpha_scale=set_alpha_scale)
        sampler.use_graph = use_graph
        if meta["inpaint"]:
            mask = draw_masks_from_boxes(batch["boxes"], hw).to(dev)
            z0 = syn.make_latent(meta.get("x0_batch", B), 4, hw, hw, seed=2).to(dev)
            extra = torch.cat([z0 * mask, mask], dim=1)
        inp = dict(x=syn.make_latent(B, 4, hw, hw, seed=6).to(dev), timesteps=None, context=ctx, grounding_input=gin,
                   inpainting_extra_input=extra, grounding_extra_input=None)
        if meta["inpaint"]:  # the S q_sample draws, as recorded from the reference run, in the reference's call order
            noise = torch.from_numpy(g["noise"]).to(dev)
            monkeypatch.setattr(torch, "randn_like", scripted_randn_like(noise, multistep=not name.startswith("ddim")))
        out = sampler.sample(S=S, shape=(B, 4, hw, hw), input=inp, uc=uc, guidance_scale=7.5, mask=mask, x0=z0)
        monkeypatch.undo()
        monkeypatch.chdir(tmp_path)
        results.append(out.clone())
        model._drop_engine()
    rel = mse(results[0], g["x_out"]) / float(g["x_out"].var())
    REPORT[name] = dict(x_rel_mse=rel, x_std=float(g["x_out"].std()))
    # bf16 engine vs fp32 reference through S chained CFG evaluations of a random-weight UNet
    assert rel < PLMS_TOL, REPORT[name]
    assert torch.equal(results[0], results[1]), "hipGraph replay must reproduce the eager launch sequence bit for bit"


def test_smoke_entry():
    import __graft_entry__ as ge
    ge.smoke()


def test_ff_rows_policy_modes_and_box_calibration():
    """The engine's choice between the row-local feed-forward kernel and LayerNorm + two GEMMs (gl_set_ff_rows_policy): never / wherever it
    exists / decided by on-device timing (the default) all reproduce the reference's eps of the small UNet (its C = 320 level at
    B x 16 x 16 = 512 rows satisfies the kernel's M % 128 == 0), the timed mode records what it measured, and gl_box_calibrate returns
    the three box numbers in physically possible ranges."""
    dev = _dev()
    g = load_golden("unet_small_text")
    meta = g["meta"]
    model = build_product_unet(meta["cfg"], meta["kind"], meta["inpaint"], device=dev)
    batch, x, ctx, t, extra = unet_inputs(meta)
    gin = model.grounding_tokenizer_input.prepare(_to(batch, dev))
    inp = dict(x=x.to(dev), timesteps=t.to(dev), context=ctx.to(dev), grounding_input=gin, inpainting_extra_input=None, grounding_extra_input=None)
    eng = model.engine
    out = {}
    try:
        for mode in (0, 1, -1):
            eng.set_ff_rows_policy(mode)
            out[mode] = model(inp)
            assert mse(out[mode], g["eps"]) < EPS_MSE_TOL, (mode, mse(out[mode], g["eps"]))
            assert torch.equal(out[mode], model(inp))        # a decided shape stays decided: same kernels, same bits
    finally:
        eng.set_ff_rows_policy(-1)
    assert not torch.equal(out[0], out[1])                   # the two forms are different kernels (bf16 summation order)
    rep = eng.ff_rows_policy_report()
    assert rep.startswith("mode=-1") and "C320 M512" in rep and ("-> rows" in rep or "-> gemm" in rep), rep
    assert torch.equal(out[-1], out[1]) or torch.equal(out[-1], out[0])
    REPORT["ff_rows_policy"] = rep
    cal = eng.box_calibrate()
    REPORT["box_calibration"] = cal
    assert 500 < cal["hbm_copy_GBps"] < 8000 and 1 < cal["lds_dma_TBps"] < 60 and 300 < cal["mfma_bf16_TFLOPs"] < 2600, cal
