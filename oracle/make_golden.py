"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (read-only import from /root/reference).

Test infrastructure. Run in the build container only (the GPU box has no /root/reference):

    cd /tmp && python /root/repo/oracle/make_golden.py [--only NAME ...]

The reference's `ldm` / `grounding_input` packages collide by name with this repo's drop-in
packages, so this script puts /root/reference first on sys.path and loads this repo's
gligen_amd/synthetic.py by file path. Weights are never stored: both sides regenerate them from
the per-key seeds in gligen_amd.synthetic; only inputs' seeds and the reference's outputs are saved.
"""
import argparse
import ast
import importlib.util
import json
import os
import sys
import time

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402
import torch  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


syn = _load("gl_synthetic", os.path.join(REPO, "gligen_amd", "synthetic.py"))

from ldm.models.autoencoder import AutoencoderKL  # noqa: E402  (reference)
from ldm.models.diffusion.ldm import LatentDiffusion  # noqa: E402
from ldm.models.diffusion.plms import PLMSSampler  # noqa: E402
from ldm.models.diffusion.ddim import DDIMSampler  # noqa: E402
from ldm.modules.attention import GatedCrossAttentionDense, GatedSelfAttentionDense  # noqa: E402
from ldm.modules.diffusionmodules.openaimodel import UNetModel  # noqa: E402
from ldm.util import instantiate_from_config  # noqa: E402

assert sys.modules["ldm.util"].__file__.startswith(REF), "must import the reference's ldm package"
OUT = os.path.join(REPO, "tests", "golden")
GINPUT = {
    "text": "grounding_input.text_grounding_tokinzer_input.GroundingNetInput",
    "text_image": "grounding_input.text_image_grounding_tokinzer_input.GroundingNetInput",
    "keypoint": "grounding_input.keypoint_grounding_tokinzer_input.GroundingNetInput",
}


def _extract_function(path, name):
    """Pull one pure function out of a reference script whose module-level imports are unavailable."""
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            ns = {"np": np, "torch": torch, "random": __import__("random")}
            exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
            return ns[name]
    raise KeyError(name)


ref_alpha_generator = _extract_function(os.path.join(REF, "gligen_inference.py"), "alpha_generator")
ref_draw_masks = _extract_function(os.path.join(REF, "inpaint_mask_func.py"), "draw_masks_from_boxes")


def set_alpha_scale(model, alpha_scale):  # reference gligen_inference.py:24-28
    for module in model.modules():
        if type(module) == GatedCrossAttentionDense or type(module) == GatedSelfAttentionDense:
            module.scale = alpha_scale


def build_unet(cfg, kind, inpaint=False, seed=1234):
    params = dict(cfg, grounding_tokenizer=syn.GROUNDING_TOKENIZERS[kind], inpaint_mode=inpaint)
    model = UNetModel(**params).eval()
    syn.fill_module_(model, seed)
    model.grounding_tokenizer_input = instantiate_from_config(dict(target=GINPUT[kind]))
    return model


def unet_case(name, cfg, kind, B, hw, inpaint=False, n_valid=3, max_objs=30):
    t0 = time.time()
    model = build_unet(cfg, kind, inpaint)
    batch = syn.make_batch(kind, B, n_valid=n_valid, seed=1, max_objs=max_objs)
    g = model.grounding_tokenizer_input.prepare(batch)
    x = syn.make_latent(B, 4, hw, hw, seed=1)
    ctx = syn.make_context(B, seed=1)
    t = torch.tensor([981, 441][:B] if B <= 2 else [981] * B, dtype=torch.long)
    extra = None
    if inpaint:
        mask = ref_draw_masks(batch["boxes"], hw)
        z0 = syn.make_latent(B, 4, hw, hw, seed=2)
        extra = torch.cat([z0 * mask, mask], dim=1)
    inp = dict(x=x, timesteps=t, context=ctx, grounding_input=g, inpainting_extra_input=extra, grounding_extra_input=None)
    out = {}
    with torch.no_grad():
        out["eps"] = model(inp).numpy()
        inp_null = {k: v for k, v in inp.items() if k != "grounding_input"}
        out["eps_null"] = model(inp_null).numpy()
        set_alpha_scale(model, 0.3)
        out["eps_scale03"] = model(inp).numpy()
        set_alpha_scale(model, 1)
        out["objs"] = model.position_net(**g).numpy()
    meta = dict(cfg=cfg, kind=kind, B=B, hw=hw, inpaint=inpaint, n_valid=n_valid, max_objs=max_objs, weight_seed=1234,
                n_keys=len(model.state_dict()), n_params=int(sum(p.numel() for p in model.parameters())))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=json.dumps(meta), **out)
    shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
    print(f"{name}: eps std {out['eps'].std():.4f} mean {out['eps'].mean():.4f}; cond-vs-null mse "
          f"{((out['eps'] - out['eps_null']) ** 2).mean():.3e}; scale mse {((out['eps'] - out['eps_scale03']) ** 2).mean():.3e} "
          f"[{time.time() - t0:.1f}s]")
    return shapes


def vae_case(name, dd, B, hw):
    t0 = time.time()
    ae = AutoencoderKL(ddconfig=dd, embed_dim=4, scale_factor=0.18215).eval()
    syn.fill_module_(ae, 4321)
    z = syn.make_latent(B, 4, hw, hw, seed=3) * 0.18215 * 4
    with torch.no_grad():
        img = ae.decode(z).numpy()
    meta = dict(ddconfig=dd, B=B, hw=hw, weight_seed=4321, n_keys=len(ae.state_dict()))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=json.dumps(meta), img=img)
    print(f"{name}: img std {img.std():.4f} min {img.min():.3f} max {img.max():.3f} [{time.time() - t0:.1f}s]")
    return {k: list(v.shape) for k, v in ae.state_dict().items()}


def vae_encode_case(name, dd, B, res):
    """AutoencoderKL.encode of the reference on a seeded image; the posterior's torch.randn draw is recorded."""
    t0 = time.time()
    ae = AutoencoderKL(ddconfig=dd, embed_dim=4, scale_factor=0.18215).eval()
    syn.fill_module_(ae, 4321)
    x = torch.rand(B, 3, res, res, generator=torch.Generator().manual_seed(8)) * 2 - 1
    draws = []
    real_randn = torch.randn

    def rec_randn(*a, **k):
        t = real_randn(*a, generator=torch.Generator().manual_seed(99), **k)
        draws.append(t)
        return t

    torch.randn = rec_randn
    try:
        with torch.no_grad():
            z = ae.encode(x)
    finally:
        torch.randn = real_randn
    assert len(draws) == 1
    meta = dict(ddconfig=dd, B=B, res=res, weight_seed=4321)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=json.dumps(meta), z=z.numpy(), noise=draws[0].numpy())
    print(f"{name}: z {tuple(z.shape)} std {z.std():.4f} [{time.time() - t0:.1f}s]")


class Recorder(torch.nn.Module):
    """Mock UNet that records the call sequence (SURVEY.md §3.1 probe) and returns a cheap function of x."""

    def __init__(self):
        super().__init__()
        self.fuser = GatedSelfAttentionDense(8, 8, 1, 8)
        self.calls = []
        self.restores = 0

    def restore_first_conv_from_SD(self):
        self.restores += 1

    def forward(self, inp):
        self.calls.append((int(inp["timesteps"][0]), "grounding_input" in inp, float(self.fuser.scale)))
        return torch.tanh(inp["x"]) * (0.5 if "grounding_input" in inp else 0.3) + 0.01 * inp["timesteps"].float().view(-1, 1, 1, 1) / 1000


def plms_trace_case(name, S, alpha_type, sampler_cls=None):
    diffusion = LatentDiffusion(linear_start=0.00085, linear_end=0.012, timesteps=1000)
    mock = Recorder()
    from functools import partial
    sampler_cls = sampler_cls or PLMSSampler
    sampler = sampler_cls(diffusion, mock, alpha_generator_func=partial(ref_alpha_generator, type=alpha_type), set_alpha_scale=set_alpha_scale)
    x = syn.make_latent(2, 4, 8, 8, seed=5)
    inp = dict(x=x.clone(), timesteps=None, context=torch.zeros(2, 1, 1), grounding_input={}, inpainting_extra_input=None, grounding_extra_input=None)
    out = sampler.sample(S=S, shape=(2, 4, 8, 8), input=inp, uc=torch.ones(2, 1, 1), guidance_scale=7.5)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), x_out=out.numpy(), calls=np.asarray(mock.calls, dtype=np.float64),
                        restores=mock.restores, ddim_timesteps=sampler.ddim_timesteps, ddim_alphas=np.asarray(sampler.ddim_alphas),
                        ddim_alphas_prev=np.asarray(sampler.ddim_alphas_prev),
                        meta=json.dumps(dict(S=S, alpha_type=alpha_type, guidance_scale=7.5)))
    print(f"{name}: {len(mock.calls)} model calls, {mock.restores} first-conv restores")


def plms_unet_case(name, S, hw, alpha_type, inpaint=False, sampler_cls=None, cfg=None, max_objs=30, B=2, x0_batch=None):
    """x0_batch=1: one encoded input image broadcast over a larger latent batch, as gligen_inference.run() does
    (reference gligen_inference.py:396-407 with the CLI's default batch_size 5)."""
    t0 = time.time()
    from functools import partial
    sampler_cls = sampler_cls or PLMSSampler
    diffusion = LatentDiffusion(linear_start=0.00085, linear_end=0.012, timesteps=1000)
    model = build_unet(cfg or syn.UNET_CFG_SMALL, "text", inpaint)
    # restore_first_conv_from_SD th.load()s a cwd-relative file: give it a seeded stand-in for the real
    # SD weights (same shapes), so the test can rebuild the identical file without /root/reference
    import tempfile
    tmp = tempfile.mkdtemp()
    torch.save(syn.sd_first_conv_state(), os.path.join(tmp, "SD_input_conv_weight_bias.pth"))
    os.chdir(tmp)
    batch = syn.make_batch("text", B, n_valid=3, seed=1, max_objs=max_objs)
    g = model.grounding_tokenizer_input.prepare(batch)
    x = syn.make_latent(B, 4, hw, hw, seed=6)
    ctx, uc = syn.make_context(B, seed=1), syn.make_context(B, seed=9)
    mask = z0 = extra = None
    noise = None
    if inpaint:
        mask = ref_draw_masks(batch["boxes"], hw)
        z0 = syn.make_latent(x0_batch or B, 4, hw, hw, seed=2)
        extra = torch.cat([z0 * mask, mask], dim=1)
        noise = torch.randn(S, x0_batch or B, 4, hw, hw, generator=torch.Generator().manual_seed(77))
        draws = iter(noise)
        diffusion_q = diffusion.q_sample
        diffusion.q_sample = lambda x_start, t, noise=None: diffusion_q(x_start, t, noise=next(draws))
    sampler = sampler_cls(diffusion, model, alpha_generator_func=partial(ref_alpha_generator, type=alpha_type), set_alpha_scale=set_alpha_scale)
    inp = dict(x=x.clone(), timesteps=None, context=ctx, grounding_input=g, inpainting_extra_input=extra, grounding_extra_input=None)
    with torch.no_grad():
        out = sampler.sample(S=S, shape=tuple(x.shape), input=inp, uc=uc, guidance_scale=7.5, mask=mask, x0=z0)
    extra_out = {} if noise is None else dict(noise=noise.numpy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), x_out=out.numpy(), **extra_out,
                        meta=json.dumps(dict(S=S, hw=hw, alpha_type=alpha_type, guidance_scale=7.5, B=B, inpaint=inpaint, n_valid=3,
                                             cfg=cfg or syn.UNET_CFG_SMALL, max_objs=max_objs, x0_batch=x0_batch or B)))
    print(f"{name}: x_out std {out.std():.4f} [{time.time() - t0:.1f}s]")


def unet_pair_case(name, kind, B, hw, n_valid=8, inpaint=False):
    """The shipped topology at the benchmark's latent size: eps of the grounded batch and of the null-grounding / uc batch
    (the two halves of the engine's [cond ; uncond] evaluation, reference plms.py:116-122). inpaint: the 9-channel first
    conv of BASELINE C4 with the masked latent + mask of gligen_inference.py:396-407 (the same tensor in both halves)."""
    t0 = time.time()
    model = build_unet(syn.UNET_CFG, kind, inpaint)
    batch = syn.make_batch(kind, B, n_valid=n_valid, seed=3)
    g = model.grounding_tokenizer_input.prepare(batch)
    x = syn.make_latent(B, 4, hw, hw, seed=3)
    ctx, uc = syn.make_context(B, seed=3), syn.make_context(B, seed=9)
    t = torch.full((B,), 501, dtype=torch.long)
    extra = None
    if inpaint:
        mask = ref_draw_masks(batch["boxes"], hw)
        z0 = syn.make_latent(B, 4, hw, hw, seed=2)
        extra = torch.cat([z0 * mask, mask], dim=1)
    with torch.no_grad():
        eps = model(dict(x=x, timesteps=t, context=ctx, grounding_input=g, inpainting_extra_input=extra, grounding_extra_input=None)).numpy()
        eps_u = model(dict(x=x, timesteps=t, context=uc, inpainting_extra_input=extra, grounding_extra_input=None)).numpy()
        objs = model.position_net(**g).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), eps=eps.astype(np.float16), eps_uncond=eps_u.astype(np.float16), objs=objs.astype(np.float16),
                        meta=json.dumps(dict(kind=kind, B=B, hw=hw, n_valid=n_valid, t=501, stored="float16", inpaint=inpaint)))
    print(f"{name}: eps std {eps.std():.4f} cond-vs-uncond mse {((eps - eps_u) ** 2).mean():.3e} [{time.time() - t0:.1f}s]")


def _timm_shim():
    """The reference's convnext.py imports timm (absent here) for trunc_normal_ (init only), DropPath (identity at rate 0)
    and the register_model decorator; none of them takes part in the forward that is recorded."""
    import types
    if "timm" in sys.modules:
        return
    timm, models, layers, registry = (types.ModuleType(n) for n in ("timm", "timm.models", "timm.models.layers", "timm.models.registry"))
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    layers.DropPath = torch.nn.Identity
    registry.register_model = lambda f: f
    timm.models, models.layers, models.registry = models, layers, registry
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers, "timm.models.registry": registry})


SPATIAL_KEYS = dict(canny="canny_edge", hed="hed_edge", depth="depth", normal="normal", sem="sem")


def spatial_case(name, modality, B=2, hw=16, res=128):
    """A spatial-map modality end to end in the reference: GroundingDownsampler -> 4 + k channel first conv, ConvNeXt-tiny
    tokenizer -> gatedSA fusers (configs/cc3m_canny.yaml etc. on the small UNet). The conditioning map is a seeded random
    image at res x res (the modules resize it themselves)."""
    t0 = time.time()
    _timm_shim()
    key = SPATIAL_KEYS[modality]
    ds_params = dict(out_dim=1) if modality == "hed" else dict(resize_input=4 * hw, out_dim=8)
    tk_params = dict(resize_input=128, out_dim=768)
    if modality == "sem":
        ds_params["in_dim"], tk_params["in_dim"] = 152, 152
    cfg = dict(syn.UNET_CFG_SMALL,
               grounding_downsampler=dict(target=f"ldm.modules.diffusionmodules.{modality}_grounding_downsampler.GroundingDownsampler", params=ds_params),
               grounding_tokenizer=dict(target=f"ldm.modules.diffusionmodules.{modality}_grounding_net.PositionNet", params=tk_params))
    import ldm.modules.diffusionmodules.convnext as cnx
    real_hub = torch.hub.load_state_dict_from_url
    torch.hub.load_state_dict_from_url = lambda *a, **k: {"model": {}}   # pretrained=True would download ImageNet weights
    try:
        model = UNetModel(**cfg).eval()
    finally:
        torch.hub.load_state_dict_from_url = real_hub
    syn.fill_module_(model, 1234)
    gin = instantiate_from_config(dict(target=f"grounding_input.{modality}_grounding_tokinzer_input.GroundingNetInput"))
    dsin = instantiate_from_config(dict(target=f"grounding_input.{modality}_grounding_downsampler_input.GroundingDSInput"))
    model.grounding_tokenizer_input = gin
    img = syn.make_spatial_map(modality, B, res, seed=1)
    batch = {key: img, "mask": torch.ones(B, 1)}
    g = gin.prepare(batch)
    extra = dsin.prepare(batch)
    x, ctx = syn.make_latent(B, 4, hw, hw, seed=1), syn.make_context(B, seed=1)
    t = torch.tensor([981, 441][:B], dtype=torch.long)
    inp = dict(x=x, timesteps=t, context=ctx, grounding_input=g, inpainting_extra_input=None, grounding_extra_input=extra)
    out = {}
    with torch.no_grad():
        out["eps"] = model(inp).numpy()
        out["eps_null"] = model({k: v for k, v in inp.items() if k != "grounding_input"}).numpy()
        out["objs"] = model.position_net(**g).numpy().astype(np.float16)
        out["objs_null"] = model.position_net(**gin.get_null_input()).numpy().astype(np.float16)
        out["ds"] = model.downsample_net(extra).numpy()
    meta = dict(cfg=cfg, modality=modality, B=B, hw=hw, res=res, weight_seed=1234, n_keys=len(model.state_dict()))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=json.dumps(meta), **out)
    print(f"{name}: eps std {out['eps'].std():.4f}; cond-vs-null mse {((out['eps'] - out['eps_null']) ** 2).mean():.3e}; ds std {out['ds'].std():.4f} "
          f"[{time.time() - t0:.1f}s]")
    return {k: list(v.shape) for k, v in model.state_dict().items()}


def plms_spatial_case(name, modality, S=5, hw=16, res=128, alpha_type=(0.6, 0.0, 0.4), sampler_cls=None):
    """The reference's sampler on a spatial-map model (GroundingDownsampler feeding a 4 + k channel first conv): CFG pairs with
    the same grounding_extra_input in both halves (plms.py:118), an alpha schedule with gated-off steps, i.e. the SD first
    conv swapped in mid-run, after which the downsampled map is no longer concatenated (openaimodel.py:442-444)."""
    t0 = time.time()
    from functools import partial
    import tempfile
    _timm_shim()
    key = SPATIAL_KEYS[modality]
    ds_params = dict(out_dim=1) if modality == "hed" else dict(resize_input=4 * hw, out_dim=8)
    tk_params = dict(resize_input=128, out_dim=768)
    cfg = dict(syn.UNET_CFG_SMALL,
               grounding_downsampler=dict(target=f"ldm.modules.diffusionmodules.{modality}_grounding_downsampler.GroundingDownsampler", params=ds_params),
               grounding_tokenizer=dict(target=f"ldm.modules.diffusionmodules.{modality}_grounding_net.PositionNet", params=tk_params))
    real_hub = torch.hub.load_state_dict_from_url
    torch.hub.load_state_dict_from_url = lambda *a, **k: {"model": {}}
    try:
        model = UNetModel(**cfg).eval()
    finally:
        torch.hub.load_state_dict_from_url = real_hub
    syn.fill_module_(model, 1234)
    gin = instantiate_from_config(dict(target=f"grounding_input.{modality}_grounding_tokinzer_input.GroundingNetInput"))
    dsin = instantiate_from_config(dict(target=f"grounding_input.{modality}_grounding_downsampler_input.GroundingDSInput"))
    model.grounding_tokenizer_input = gin
    tmp = tempfile.mkdtemp()
    torch.save(syn.sd_first_conv_state(), os.path.join(tmp, "SD_input_conv_weight_bias.pth"))
    os.chdir(tmp)
    B = 2
    img = syn.make_spatial_map(modality, B, res, seed=1)
    batch = {key: img, "mask": torch.ones(B, 1)}
    g = gin.prepare(batch)
    extra = dsin.prepare(batch)
    x = syn.make_latent(B, 4, hw, hw, seed=6)
    ctx, uc = syn.make_context(B, seed=1), syn.make_context(B, seed=9)
    diffusion = LatentDiffusion(linear_start=0.00085, linear_end=0.012, timesteps=1000)
    sampler = (sampler_cls or PLMSSampler)(diffusion, model, alpha_generator_func=partial(ref_alpha_generator, type=list(alpha_type)),
                                           set_alpha_scale=set_alpha_scale)
    inp = dict(x=x.clone(), timesteps=None, context=ctx, grounding_input=g, inpainting_extra_input=None, grounding_extra_input=extra)
    with torch.no_grad():
        out = sampler.sample(S=S, shape=tuple(x.shape), input=inp, uc=uc, guidance_scale=7.5)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), x_out=out.numpy(),
                        meta=json.dumps(dict(S=S, hw=hw, res=res, modality=modality, alpha_type=list(alpha_type), guidance_scale=7.5, B=B, cfg=cfg,
                                             first_conv_type=model.first_conv_type)))
    print(f"{name}: x_out std {out.std():.4f} first conv now {model.first_conv_type} [{time.time() - t0:.1f}s]")


def c2_case(name="c2_end_to_end", S=50, hw=64, kind="text"):
    """BASELINE config C2 for ONE image at its real size: box+text, 8 boxes, 512x512, 50 PLMS steps (102 UNet forwards), CFG
    7.5, gate on at every step (alpha_type None = [1, 0, 0], the schedule the metric is quoted on), B = 1, fp32 on the CPU
    through the reference's PLMSSampler + UNetModel + AutoencoderKL.decode (gligen_inference.py:389-446). The latent after
    10 and 25 steps is recorded too (p_sample_plms hooked), so a divergence can be located.
    kind = "text_image" / "keypoint" (round 6): the same run for BASELINE C3 (CLIP image tokens next to the phrase tokens, Ng = 60)
    and C5 (17 COCO keypoints per person -> Fourier tokens, Ng = 136) -- same seeds, the tokenizer and its batch are what change."""
    t0 = time.time()
    from functools import partial
    diffusion = LatentDiffusion(linear_start=0.00085, linear_end=0.012, timesteps=1000)
    model = build_unet(syn.UNET_CFG, kind)
    ae = AutoencoderKL(ddconfig=syn.VAE_DDCONFIG, embed_dim=4, scale_factor=0.18215).eval()
    syn.fill_module_(ae, 4321)
    batch = syn.make_batch(kind, 1, n_valid=8, seed=1)
    g = model.grounding_tokenizer_input.prepare(batch)
    x = syn.make_latent(1, 4, hw, hw, seed=6)
    ctx, uc = syn.make_context(1, seed=1), syn.make_context(1, seed=9)
    sampler = PLMSSampler(diffusion, model, alpha_generator_func=partial(ref_alpha_generator, type=None), set_alpha_scale=set_alpha_scale)
    trace = {}
    real_step = sampler.p_sample_plms
    count = [0]

    def rec_step(*a, **k):
        r = real_step(*a, **k)
        count[0] += 1
        if count[0] in (10, 25):
            trace[f"z_step{count[0]}"] = r[0].numpy().copy()
        return r

    sampler.p_sample_plms = rec_step
    inp = dict(x=x.clone(), timesteps=None, context=ctx, grounding_input=g, inpainting_extra_input=None, grounding_extra_input=None)
    with torch.no_grad():
        z = sampler.sample(S=S, shape=tuple(x.shape), input=inp, uc=uc, guidance_scale=7.5)
        t_s = time.time() - t0
        img = ae.decode(z)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), z=z.numpy(), img=img.numpy().astype(np.float16), **trace,
                        meta=json.dumps(dict(S=S, hw=hw, alpha_type=None, guidance_scale=7.5, B=1, n_valid=8, img_stored="float16", kind=kind,
                                             ref_cpu_seconds=round(time.time() - t0, 1), ref_sampler_seconds=round(t_s, 1),
                                             cpu_threads=torch.get_num_threads())))
    print(f"{name}: z std {z.std():.4f} img std {img.std():.4f} [{time.time() - t0:.1f}s]")


def c4_case(name="c4_end_to_end", S=50, hw=64):
    """BASELINE config C4 for ONE image at its real size: inpainting box+text -- AutoencoderKL.encode of a 512x512 input image,
    mask from the boxes, 9-channel first conv on [x ; z0 * mask ; mask], per-step q_sample blend (plms.py:96-100), 50 PLMS steps
    (102 UNet forwards), CFG 7.5, decode (gligen_inference.py:396-446). fp32 on the CPU through the reference's own modules. The S
    q_sample draws are torch.randn(S, 1, 4, hw, hw) from generator seed 77 (the test regenerates them; their sum is stored as a
    check), the encoder's posterior draw from seed 5."""
    t0 = time.time()
    from functools import partial
    diffusion = LatentDiffusion(linear_start=0.00085, linear_end=0.012, timesteps=1000)
    model = build_unet(syn.UNET_CFG, "text", inpaint=True)
    ae = AutoencoderKL(ddconfig=syn.VAE_DDCONFIG, embed_dim=4, scale_factor=0.18215).eval()
    syn.fill_module_(ae, 4321)
    batch = syn.make_batch("text", 1, n_valid=8, seed=1)
    g = model.grounding_tokenizer_input.prepare(batch)
    x = syn.make_latent(1, 4, hw, hw, seed=6)
    ctx, uc = syn.make_context(1, seed=1), syn.make_context(1, seed=9)
    image = torch.rand(1, 3, 8 * hw, 8 * hw, generator=torch.Generator().manual_seed(8)) * 2 - 1
    post_noise = torch.randn(1, 4, hw, hw, generator=torch.Generator().manual_seed(5))
    real_randn = torch.randn
    torch.randn = lambda *a, **k: post_noise.clone()          # the posterior's torch.randn(mean.shape) (distributions.py:35)
    try:
        with torch.no_grad():
            z0 = ae.encode(image)
    finally:
        torch.randn = real_randn
    mask = ref_draw_masks(batch["boxes"], hw)
    extra = torch.cat([z0 * mask, mask], dim=1)
    noise = torch.randn(S, 1, 4, hw, hw, generator=torch.Generator().manual_seed(77))
    draws = iter(noise)
    diffusion_q = diffusion.q_sample
    diffusion.q_sample = lambda x_start, t, noise=None: diffusion_q(x_start, t, noise=next(draws))
    sampler = PLMSSampler(diffusion, model, alpha_generator_func=partial(ref_alpha_generator, type=None), set_alpha_scale=set_alpha_scale)
    inp = dict(x=x.clone(), timesteps=None, context=ctx, grounding_input=g, inpainting_extra_input=extra, grounding_extra_input=None)
    with torch.no_grad():
        z = sampler.sample(S=S, shape=tuple(x.shape), input=inp, uc=uc, guidance_scale=7.5, mask=mask, x0=z0)
        t_s = time.time() - t0
        img = ae.decode(z)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), z=z.numpy(), z0=z0.numpy(), img=img.numpy().astype(np.float16),
                        meta=json.dumps(dict(S=S, hw=hw, alpha_type=None, guidance_scale=7.5, B=1, n_valid=8, img_stored="float16", inpaint=True,
                                             noise_seed=77, noise_sum=float(noise.double().sum()), posterior_seed=5, image_seed=8,
                                             ref_cpu_seconds=round(time.time() - t0, 1), ref_sampler_seconds=round(t_s, 1),
                                             cpu_threads=torch.get_num_threads())))
    print(f"{name}: z std {z.std():.4f} img std {img.std():.4f} [{time.time() - t0:.1f}s]")


def c2_b4_case(name="c2_end_to_end_b4", S=50, hw=64, B=4):
    """BASELINE config C2 at the batch the metric is quoted on (B = 4 prompts with different boxes / embeddings / contexts / noise):
    50 PLMS steps, CFG 7.5, gate on at every step, fp32 on the CPU through the reference's PLMSSampler + UNetModel +
    AutoencoderKL.decode. Stored: the final latents and the decoded images average-pooled 4 x 4 (the full-resolution decode is
    pinned by the B = 1 case)."""
    t0 = time.time()
    from functools import partial
    diffusion = LatentDiffusion(linear_start=0.00085, linear_end=0.012, timesteps=1000)
    model = build_unet(syn.UNET_CFG, "text")
    ae = AutoencoderKL(ddconfig=syn.VAE_DDCONFIG, embed_dim=4, scale_factor=0.18215).eval()
    syn.fill_module_(ae, 4321)
    batch = syn.make_batch("text", B, n_valid=8, seed=21)
    g = model.grounding_tokenizer_input.prepare(batch)
    x = syn.make_latent(B, 4, hw, hw, seed=26)
    ctx, uc = syn.make_context(B, seed=21), syn.make_context(1, seed=9).expand(B, -1, -1).contiguous()
    sampler = PLMSSampler(diffusion, model, alpha_generator_func=partial(ref_alpha_generator, type=None), set_alpha_scale=set_alpha_scale)
    inp = dict(x=x.clone(), timesteps=None, context=ctx, grounding_input=g, inpainting_extra_input=None, grounding_extra_input=None)
    with torch.no_grad():
        z = sampler.sample(S=S, shape=tuple(x.shape), input=inp, uc=uc, guidance_scale=7.5)
        t_s = time.time() - t0
        img = torch.nn.functional.avg_pool2d(ae.decode(z), 4)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), z=z.numpy(), img_pool4=img.numpy().astype(np.float16),
                        meta=json.dumps(dict(S=S, hw=hw, alpha_type=None, guidance_scale=7.5, B=B, n_valid=8, img_stored="float16, avg_pool2d(4)",
                                             ref_cpu_seconds=round(time.time() - t0, 1), ref_sampler_seconds=round(t_s, 1),
                                             cpu_threads=torch.get_num_threads())))
    print(f"{name}: z std {z.std():.4f} img std {img.std():.4f} [{time.time() - t0:.1f}s]")


def c1_case(name="c1_end_to_end", S=20, hw=32):
    """BASELINE config C1: box+text, 1 box, 256x256, 20 PLMS steps, CFG 7.5, B=1, fp32 on the CPU through the reference's
    PLMSSampler + UNetModel + AutoencoderKL.decode (gligen_inference.py:343-446 with steps / image_size overridden), the first
    meta_list entry's alpha schedule [0.3, 0, 0.7] (gligen_inference.py:469-476) incl. the SD first-conv swap."""
    t0 = time.time()
    from functools import partial
    import tempfile
    diffusion = LatentDiffusion(linear_start=0.00085, linear_end=0.012, timesteps=1000)
    model = build_unet(syn.UNET_CFG, "text")
    ae = AutoencoderKL(ddconfig=syn.VAE_DDCONFIG, embed_dim=4, scale_factor=0.18215).eval()
    syn.fill_module_(ae, 4321)
    tmp = tempfile.mkdtemp()
    torch.save(syn.sd_first_conv_state(), os.path.join(tmp, "SD_input_conv_weight_bias.pth"))
    os.chdir(tmp)
    batch = syn.make_batch("text", 1, n_valid=1, seed=1)
    g = model.grounding_tokenizer_input.prepare(batch)
    x = syn.make_latent(1, 4, hw, hw, seed=6)
    ctx, uc = syn.make_context(1, seed=1), syn.make_context(1, seed=9)
    alpha_type = [0.3, 0.0, 0.7]
    sampler = PLMSSampler(diffusion, model, alpha_generator_func=partial(ref_alpha_generator, type=alpha_type), set_alpha_scale=set_alpha_scale)
    inp = dict(x=x.clone(), timesteps=None, context=ctx, grounding_input=g, inpainting_extra_input=None, grounding_extra_input=None)
    with torch.no_grad():
        z = sampler.sample(S=S, shape=tuple(x.shape), input=inp, uc=uc, guidance_scale=7.5)
        t_s = time.time() - t0
        img = ae.decode(z)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), z=z.numpy(), img=img.numpy().astype(np.float16),
                        meta=json.dumps(dict(S=S, hw=hw, alpha_type=alpha_type, guidance_scale=7.5, B=1, n_valid=1, img_stored="float16",
                                             ref_cpu_seconds=round(time.time() - t0, 1), ref_sampler_seconds=round(t_s, 1),
                                             cpu_threads=torch.get_num_threads())))
    print(f"{name}: z std {z.std():.4f} img std {img.std():.4f} [{time.time() - t0:.1f}s]")


def meta_list_case():
    """The reference's demo prompt list (gligen_inference.py:466-637), parsed out of its __main__ block."""
    tree = ast.parse(open(os.path.join(REF, "gligen_inference.py")).read())
    found = None
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == "meta_list" for t in node.targets):
            # a list of dict(...) calls over literals: evaluated with `dict` as the only visible name
            found = eval(compile(ast.Expression(node.value), "meta_list", "eval"), {"__builtins__": {}, "dict": dict})
    assert found
    json.dump(found, open(os.path.join(OUT, "meta_list.json"), "w"), indent=1)
    print(f"meta_list: {len(found)} entries")


def misc_case():
    out = {}
    for S in (20, 50):
        for tp in (None, [0.3, 0.0, 0.7], [0.5, 0.25, 0.25]):
            out[f"alpha_{S}_{'none' if tp is None else '_'.join(str(v) for v in tp)}"] = np.asarray(ref_alpha_generator(S, tp), dtype=np.float64)
    boxes, _ = syn.make_boxes(2, 5, seed=3)
    out["mask64"] = ref_draw_masks(boxes, 64).numpy()
    out["mask_boxes"] = boxes.numpy()
    diffusion = LatentDiffusion(linear_start=0.00085, linear_end=0.012, timesteps=1000)
    for k, v in diffusion.state_dict().items():
        out["diff_" + k] = v.numpy()
    from ldm.modules.diffusionmodules.util import timestep_embedding
    out["temb"] = timestep_embedding(torch.tensor([1, 441, 981]), 320).numpy()
    np.savez_compressed(os.path.join(OUT, "misc.npz"), **out)
    print("misc: ok")


def block_backward_case(name="block_backward_gatedsa", B=2, hw=16, Ng=30, C=320, heads=8, ctx_dim=768, ctx_T=77):
    """Training slice (second half): gradients of the reference's loss -- mse_loss(model_output, noise),
    trainer.py:353-371 -- through ONE BasicTransformerBlock (attention.py:333-338) from the reference's own autograd: with
    respect to the block's input, the grounding tokens and every trainable (fuser.*) parameter of trainer.py:217-245. The
    block's output plays the role of model_output, `target` of the noise."""
    from ldm.modules.attention import BasicTransformerBlock
    blk = BasicTransformerBlock(C, ctx_dim, ctx_dim, heads, C // heads, "gatedSA", use_checkpoint=False)
    syn.fill_module_(blk, 77)
    with torch.no_grad():   # the gates start at tanh(0) = 0 (attention.py:229-230), where the fuser gets no gradient but alpha: open them
        blk.fuser.alpha_attn.fill_(0.6)
        blk.fuser.alpha_dense.fill_(-0.4)
    g = torch.Generator().manual_seed(4242)
    N = hw * hw
    x = torch.randn(B, N, C, generator=g).requires_grad_(True)
    objs = (torch.randn(B, Ng, ctx_dim, generator=g) * 0.5).requires_grad_(True)
    context = torch.randn(B, ctx_T, ctx_dim, generator=g)
    target = torch.randn(B, N, C, generator=g)
    for p_name, p_ in blk.named_parameters():
        p_.requires_grad_(p_name.startswith("fuser."))
    y = blk(x, context, objs)
    loss = torch.nn.functional.mse_loss(y, target)
    loss.backward()
    # inputs are not stored: the test regenerates them from the same seeded CPU generator (block_backward_inputs in tests/helpers.py
    # repeats the four draws above); the big weight gradients are stored as fp16 of g / max|g| + the scale (2^-11 relative per entry)
    out = dict(y=y.detach().numpy(), loss=np.float64(loss.item()), dx=x.grad.numpy(), dobjs=objs.grad.numpy(),
               x_sum=np.float64(x.detach().double().sum().item()), target_sum=np.float64(target.double().sum().item()))
    for p_name, p_ in blk.named_parameters():
        if p_name.startswith("fuser."):
            gq = p_.grad.numpy()
            sc = float(np.abs(gq).max()) or 1.0
            out["grad." + p_name] = (gq / sc).astype(np.float16)
            out["scale." + p_name] = np.float64(sc)
    out["meta"] = np.frombuffer(json.dumps(dict(B=B, hw=hw, Ng=Ng, C=C, heads=heads, ctx_dim=ctx_dim, ctx_T=ctx_T, seed=77,
                                                alpha_attn=0.6, alpha_dense=-0.4)).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss {loss.item():.6f}, |dx| {x.grad.abs().mean():.3e}, {sum(1 for k in out if k.startswith('grad.'))} fuser gradients")
    return {k: list(v.shape) for k, v in blk.state_dict().items()}


def resblock_backward_case(name, Cin, Cout, B=2, hw=16, emb_dim=256):
    """Training slice, second block type: gradient of the reference's loss w.r.t. the INPUT of one ResBlock
    (openaimodel.py:154-232) from the reference's own autograd. Every ResBlock parameter is frozen in the reference's trainer
    (trainer.py:217-245), so dL/dx -- the path to the fusers in front -- is all a training step needs from it. out_layers' last
    conv is zero-initialised in the reference (zero_module); the seeded fill gives it real weights, as a trained checkpoint has."""
    from ldm.modules.diffusionmodules.openaimodel import ResBlock
    blk = ResBlock(Cin, emb_dim, 0.0, out_channels=Cout, use_checkpoint=False)
    syn.fill_module_(blk, 78)
    for p_ in blk.parameters():
        p_.requires_grad_(False)
    g = torch.Generator().manual_seed(4343)
    x = torch.randn(B, Cin, hw, hw, generator=g).requires_grad_(True)
    emb = torch.randn(B, emb_dim, generator=g)
    target = torch.randn(B, Cout, hw, hw, generator=g)
    y = blk(x, emb)
    loss = torch.nn.functional.mse_loss(y, target)
    loss.backward()
    out = dict(y=y.detach().numpy(), loss=np.float64(loss.item()), dx=x.grad.numpy(), x_sum=np.float64(x.detach().double().sum().item()),
               target_sum=np.float64(target.double().sum().item()))
    out["meta"] = np.frombuffer(json.dumps(dict(B=B, hw=hw, Cin=Cin, Cout=Cout, emb_dim=emb_dim, seed=78)).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss {loss.item():.6f}, |dx| {x.grad.abs().mean():.3e}, keys {sorted(blk.state_dict().keys())}")
    return {k: list(v.shape) for k, v in blk.state_dict().items()}


def st_backward_case(name="st_backward_gatedsa", B=2, hw=16, Ng=30, C=128, heads=4, ctx_dim=768, ctx_T=77):
    """Training slice: one SpatialTransformer (attention.py:340-376: GroupNorm, proj_in, a gatedSA BasicTransformerBlock, proj_out,
    residual) under the reference's loss, gradients from the reference's autograd for its trainable parameters
    (transformer_blocks.0.fuser.*, trainer.py:217-245), its input and the grounding tokens."""
    from ldm.modules.attention import SpatialTransformer
    st = SpatialTransformer(C, ctx_dim, ctx_dim, heads, C // heads, depth=1, fuser_type="gatedSA", use_checkpoint=False)
    syn.fill_module_(st, 79)
    with torch.no_grad():
        st.transformer_blocks[0].fuser.alpha_attn.fill_(0.5)
        st.transformer_blocks[0].fuser.alpha_dense.fill_(-0.7)
    g = torch.Generator().manual_seed(4444)
    x = torch.randn(B, C, hw, hw, generator=g).requires_grad_(True)
    objs = (torch.randn(B, Ng, ctx_dim, generator=g) * 0.5).requires_grad_(True)
    context = torch.randn(B, ctx_T, ctx_dim, generator=g)
    target = torch.randn(B, C, hw, hw, generator=g)
    for p_name, p_ in st.named_parameters():
        p_.requires_grad_(".fuser." in p_name)
    y = st(x, context, objs)
    loss = torch.nn.functional.mse_loss(y, target)
    loss.backward()
    out = dict(y=y.detach().numpy(), loss=np.float64(loss.item()), dx=x.grad.numpy(), dobjs=objs.grad.numpy(),
               x_sum=np.float64(x.detach().double().sum().item()), target_sum=np.float64(target.double().sum().item()))
    for p_name, p_ in st.named_parameters():
        if ".fuser." in p_name:
            gq = p_.grad.numpy()
            sc = float(np.abs(gq).max()) or 1.0
            out["grad." + p_name] = (gq / sc).astype(np.float16)
            out["scale." + p_name] = np.float64(sc)
    out["meta"] = np.frombuffer(json.dumps(dict(B=B, hw=hw, Ng=Ng, C=C, heads=heads, ctx_dim=ctx_dim, ctx_T=ctx_T, seed=79,
                                                alpha_attn=0.5, alpha_dense=-0.7)).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: loss {loss.item():.6f}, |dx| {x.grad.abs().mean():.3e}, {sum(1 for k in out if k.startswith('grad.'))} fuser gradients")
    return {k: list(v.shape) for k, v in st.state_dict().items()}


def resample_backward_case(name="resample_backward", B=2, hw=16, C=64):
    """Training slice: Downsample (conv3x3 stride 2, openaimodel.py:99-124) and Upsample (nearest 2x + conv3x3, :64-96): forward and the
    gradient of mse_loss w.r.t. the input from the reference's autograd (the convs are frozen SD layers)."""
    from ldm.modules.diffusionmodules.openaimodel import Downsample, Upsample
    out = {}
    g = torch.Generator().manual_seed(4545)
    for key, mod, ho in (("down", Downsample(C, True, dims=2), hw // 2), ("up", Upsample(C, True, dims=2), hw * 2)):
        syn.fill_module_(mod, 80)
        for p_ in mod.parameters():
            p_.requires_grad_(False)
        x = torch.randn(B, C, hw, hw, generator=g).requires_grad_(True)
        target = torch.randn(B, C, ho, ho, generator=g)
        y = mod(x)
        loss = torch.nn.functional.mse_loss(y, target)
        loss.backward()
        out.update({key + "_y": y.detach().numpy(), key + "_loss": np.float64(loss.item()), key + "_dx": x.grad.numpy(),
                    key + "_x_sum": np.float64(x.detach().double().sum().item())})
        print(f"{name}/{key}: loss {loss.item():.6f}, |dx| {x.grad.abs().mean():.3e}, keys {sorted(mod.state_dict().keys())}")
    out["meta"] = np.frombuffer(json.dumps(dict(B=B, hw=hw, C=C, seed=80)).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def _grad_sample(gq, n=4096):
    """A gradient tensor as the golden stores it: the whole tensor when it has at most n elements, else n elements at a fixed
    stride of the flattened tensor (the test computes the same indices); fp16 of g / max|g| + the scale, and the full L2 norm."""
    flat = gq.reshape(-1)
    stride = max(1, flat.size // n)
    sub = flat[::stride][:n] if flat.size > n else flat
    sc = float(np.abs(sub).max()) or 1.0
    return (sub / sc).astype(np.float16), sc, float(np.sqrt((flat.astype(np.float64) ** 2).sum()))


def unet_backward_case(name="unet_small_train_step", B=2, hw=16, n_valid=3, base_cfg=None, kind="text"):
    """One whole training iteration of the reference on the small UNet (trainer.py:353-392): model(input) on a noised latent,
    mse_loss(model_output, noise), loss.backward(), with requires_grad exactly as the trainer sets it (trainer.py:217-245: every
    fuser.* parameter and position_net). Stored: loss, eps, and per trainable tensor a strided sample of its gradient + its norm
    (34 M gradient values as a whole would be 70 MB)."""
    cfg = dict(base_cfg or syn.UNET_CFG_SMALL, use_checkpoint=False)
    model = build_unet(cfg, kind)
    batch = syn.make_batch(kind, B, n_valid=n_valid, seed=5)
    if kind == "text_image":     # not every box has both modalities (text_image_grounding_net.py:57-58 masks them separately)
        batch["text_masks"][:, 1] = 0
        batch["image_masks"][:, 0] = 0
    g = model.grounding_tokenizer_input.prepare(batch)
    x = syn.make_latent(B, 4, hw, hw, seed=6)
    ctx = syn.make_context(B, seed=6)
    t = torch.tensor([981, 441][:B], dtype=torch.long)
    target = syn.make_latent(B, 4, hw, hw, seed=7)
    trainable = []
    for k, p_ in model.named_parameters():
        on = ".fuser." in k or k.startswith("position_net.")
        p_.requires_grad_(on)
        if on:
            trainable.append(k)
    eps = model(dict(x=x, timesteps=t, context=ctx, grounding_input=g, inpainting_extra_input=None, grounding_extra_input=None))
    loss = torch.nn.functional.mse_loss(eps, target)
    loss.backward()
    out = dict(eps=eps.detach().numpy(), loss=np.float64(loss.item()))
    for k, p_ in model.named_parameters():
        if k in trainable:
            sub, sc, nrm = _grad_sample(p_.grad.numpy())
            out["grad." + k] = sub
            out["scale." + k] = np.float64(sc)
            out["norm." + k] = np.float64(nrm)
    meta = dict(cfg=cfg, B=B, hw=hw, n_valid=n_valid, weight_seed=1234, n_trainable=len(trainable), sample=4096, kind=kind)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=json.dumps(meta), **out)
    print(f"{name}: loss {loss.item():.6f}, {len(trainable)} trainable tensors, {sum(p.numel() for k, p in model.named_parameters() if k in trainable) / 1e6:.1f} M gradient values")
    return {k: list(v.shape) for k, v in model.state_dict().items()}


def unet_train_2steps_case(name="unet_small_train_2steps", B=2, hw=16, n_valid=3, lr=1e-3):
    """Two optimisation steps of the reference's trainer on the small UNet (trainer.py:217-245 trainable set + torch.optim.AdamW,
    :353-392 run_one_step / backward / opt.step) on one fixed batch: the three losses (before, after one, after two updates) and
    samples of two updated tensors. lr is larger than the reference's 5e-5 so that two steps move the loss visibly."""
    cfg = dict(syn.UNET_CFG_SMALL, use_checkpoint=False)
    model = build_unet(cfg, "text")
    batch = syn.make_batch("text", B, n_valid=n_valid, seed=5)
    g = model.grounding_tokenizer_input.prepare(batch)
    x, ctx = syn.make_latent(B, 4, hw, hw, seed=6), syn.make_context(B, seed=6)
    t = torch.tensor([981, 441][:B], dtype=torch.long)
    target = syn.make_latent(B, 4, hw, hw, seed=7)
    params = []
    for k, p_ in model.named_parameters():
        on = ".fuser." in k or k.startswith("position_net.")
        p_.requires_grad_(on)
        if on:
            params.append(p_)
    opt = torch.optim.AdamW(params, lr=lr, weight_decay=0.0)
    inp = dict(x=x, timesteps=t, context=ctx, grounding_input=g, inpainting_extra_input=None, grounding_extra_input=None)
    losses = []
    for it in range(3):
        loss = torch.nn.functional.mse_loss(model(inp), target)
        losses.append(loss.item())
        if it < 2:
            opt.zero_grad()
            loss.backward()
            opt.step()
    sd = model.state_dict()
    out = dict(losses=np.asarray(losses, dtype=np.float64),
               w_linear=sd["input_blocks.1.1.transformer_blocks.0.fuser.linear.weight"].numpy().reshape(-1)[::61][:4096].copy(),
               w_pn=sd["position_net.linears.4.weight"].numpy().reshape(-1)[::97][:4096].copy())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), meta=json.dumps(dict(cfg=cfg, B=B, hw=hw, n_valid=n_valid, weight_seed=1234, lr=lr)), **out)
    print(f"{name}: losses {losses}")


CASES = {
    "unet_small_text": lambda: unet_case("unet_small_text", syn.UNET_CFG_SMALL, "text", 2, 16),
    "unet_small_text_image": lambda: unet_case("unet_small_text_image", syn.UNET_CFG_SMALL, "text_image", 2, 16),
    "unet_small_keypoint": lambda: unet_case("unet_small_keypoint", syn.UNET_CFG_SMALL, "keypoint", 2, 16),
    "unet_small_inpaint": lambda: unet_case("unet_small_inpaint", syn.UNET_CFG_SMALL, "text", 2, 16, inpaint=True),
    "unet_small_gatedca": lambda: unet_case("unet_small_gatedca", dict(syn.UNET_CFG_SMALL, fuser_type="gatedCA"), "text", 2, 16),
    # gatedSA2 needs a square grid of grounding tokens (attention.py:279-283): 16 box slots = 4 x 4
    "unet_small_gatedsa2": lambda: unet_case("unet_small_gatedsa2", dict(syn.UNET_CFG_SMALL, fuser_type="gatedSA2"), "text", 2, 16,
                                             max_objs=16),
    "unet_full_text": lambda: unet_case("unet_full_text", syn.UNET_CFG, "text", 1, 16),
    "vae_small": lambda: vae_case("vae_small", syn.VAE_DDCONFIG_SMALL, 2, 16),
    "vae_full": lambda: vae_case("vae_full", syn.VAE_DDCONFIG, 1, 8),
    "plms_trace_50": lambda: plms_trace_case("plms_trace_50", 50, [0.3, 0.0, 0.7]),
    "plms_trace_20": lambda: plms_trace_case("plms_trace_20", 20, None),
    "plms_unet_small": lambda: plms_unet_case("plms_unet_small", 5, 16, [0.6, 0.0, 0.4]),
    "plms_unet_small_inpaint": lambda: plms_unet_case("plms_unet_small_inpaint", 4, 16, None, inpaint=True),
    "ddim_trace_25": lambda: plms_trace_case("ddim_trace_25", 25, [0.4, 0.2, 0.4], sampler_cls=DDIMSampler),
    "ddim_unet_small": lambda: plms_unet_case("ddim_unet_small", 6, 16, [0.5, 0.0, 0.5], sampler_cls=DDIMSampler),
    "ddim_unet_small_inpaint": lambda: plms_unet_case("ddim_unet_small_inpaint", 4, 16, None, inpaint=True, sampler_cls=DDIMSampler),
    "vae_enc_small": lambda: vae_encode_case("vae_enc_small", syn.VAE_DDCONFIG_SMALL, 2, 64),
    "vae_enc_full": lambda: vae_encode_case("vae_enc_full", syn.VAE_DDCONFIG, 1, 64),
    "misc": misc_case,
    "meta_list": meta_list_case,
    # ---- round 2: the BASELINE configurations at their real sizes, sampler-level cases for the remaining code paths
    "vae_enc_512": lambda: vae_encode_case("vae_enc_512", syn.VAE_DDCONFIG, 1, 512),
    "plms50_unet_small": lambda: plms_unet_case("plms50_unet_small", 50, 16, [0.3, 0.0, 0.7]),
    "plms_unet_small_gatedsa2": lambda: plms_unet_case("plms_unet_small_gatedsa2", 5, 16, [0.6, 0.0, 0.4],
                                                       cfg=dict(syn.UNET_CFG_SMALL, fuser_type="gatedSA2"), max_objs=16),
    "plms_unet_small_inpaint_x0b1": lambda: plms_unet_case("plms_unet_small_inpaint_x0b1", 4, 16, None, inpaint=True, B=3, x0_batch=1),
    "unet_full_64_text": lambda: unet_pair_case("unet_full_64_text", "text", 4, 64),
    "unet_full_64_text_image": lambda: unet_pair_case("unet_full_64_text_image", "text_image", 1, 64),
    "unet_full_64_keypoint": lambda: unet_pair_case("unet_full_64_keypoint", "keypoint", 1, 64),
    "c1_end_to_end": c1_case,
    "c2_end_to_end_b4": c2_b4_case,
    # spatial-map modalities
    "unet_small_canny": lambda: spatial_case("unet_small_canny", "canny"),
    "unet_small_hed": lambda: spatial_case("unet_small_hed", "hed", B=1, hw=64),  # the hed downsampler always resizes to 64 x 64
    "unet_small_normal": lambda: spatial_case("unet_small_normal", "normal"),
    "unet_small_sem": lambda: spatial_case("unet_small_sem", "sem"),
    # ---- round 3: the remaining parity holes
    "unet_small_depth": lambda: spatial_case("unet_small_depth", "depth"),
    "plms_unet_small_canny": lambda: plms_spatial_case("plms_unet_small_canny", "canny"),
    "ddim_unet_small_hed": lambda: plms_spatial_case("ddim_unet_small_hed", "hed", S=6, hw=64, alpha_type=(0.5, 0.0, 0.5), sampler_cls=DDIMSampler),
    "unet_full_64_inpaint": lambda: unet_pair_case("unet_full_64_inpaint", "text", 1, 64, inpaint=True),
    "unet_full_64_text_image_b4": lambda: unet_pair_case("unet_full_64_text_image_b4", "text_image", 4, 64),
    "unet_full_64_keypoint_b4": lambda: unet_pair_case("unet_full_64_keypoint_b4", "keypoint", 4, 64),
    "c2_end_to_end": c2_case,
    "c4_end_to_end": c4_case,
    # ---- round 6: C3 / C5 end to end (the other two tokenizers through the whole 50-step run + decode)
    "c3_end_to_end": lambda: c2_case("c3_end_to_end", kind="text_image"),
    "c5_end_to_end": lambda: c2_case("c5_end_to_end", kind="keypoint"),
    # ---- round 4: the training slice (gradients through one transformer block, from the reference's autograd)
    "block_backward_gatedsa": block_backward_case,
    "st_backward_gatedsa": st_backward_case,
    "unet_small_train_step": unet_backward_case,
    "unet_small_train_2steps": unet_train_2steps_case,
    "unet_small_ti_train_step": lambda: unet_backward_case("unet_small_ti_train_step", kind="text_image"),
    "unet_small_kp_train_step": lambda: unet_backward_case("unet_small_kp_train_step", kind="keypoint"),
    # the shipped topology (4 levels, 16 fusers, head dims 40 / 80 / 160; 966 tensors) at a 16 x 16 latent
    "unet_full_train_step": lambda: unet_backward_case("unet_full_train_step", B=1, hw=16, base_cfg=syn.UNET_CFG),
    "resample_backward": resample_backward_case,
    "resblock_backward_skipconv": lambda: resblock_backward_case("resblock_backward_skipconv", 64, 128),
    "resblock_backward_identity": lambda: resblock_backward_case("resblock_backward_identity", 128, 128),
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    shapes = {}
    for name, fn in CASES.items():
        if args.only and name not in args.only:
            continue
        r = fn()
        if isinstance(r, dict):
            shapes[name] = r
    if shapes:
        path = os.path.join(OUT, "state_dict_shapes.json")
        old = json.load(open(path)) if os.path.exists(path) else {}
        old.update(shapes)
        json.dump(old, open(path, "w"), sort_keys=True)
