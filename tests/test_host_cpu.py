"""Host-side logic and the drop-in boundary, no GPU: state_dict key contract against the reference,
C-ABI exports, plugin glue, grounding-input adapters, sampler orchestration against the reference's
recorded call trace, loud failure without a HIP device."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import GINPUT, ROOT, golden_shapes, load_golden, mse
from gligen_amd import synthetic as syn


def test_c_abi_exports_every_declared_symbol():
    from gligen_amd import _lib
    from gligen_amd.build import build_native
    build_native()
    header = open(os.path.join(ROOT, "include", "gligen_amd.h")).read()
    declared = set(re.findall(r"\b(gl_[a-z0-9_]+)\s*\(", header))
    assert declared, "header parse failed"
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/gligen_amd.h but not exported"
    assert declared == set(_lib.SYMBOLS), "ctypes table and header disagree"
    _lib.load()


def test_no_device_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gligen_amd import GligenAmdError
    from gligen_amd.engine import Engine
    with pytest.raises(GligenAmdError):
        Engine(0)
    from helpers import build_product_unet
    model = build_product_unet(syn.UNET_CFG_SMALL, "text")
    gin = model.grounding_tokenizer_input.prepare(syn.make_batch("text", 1))
    with pytest.raises(RuntimeError, match="no CPU implementation|HIP device"):
        model(dict(x=torch.zeros(1, 4, 8, 8), timesteps=torch.zeros(1, dtype=torch.long), context=torch.zeros(1, 77, 768),
                   grounding_input=gin))
    # and the library itself refuses to create a context
    from gligen_amd import _lib
    lib = _lib.load()
    ctx = ctypes.c_void_p()
    assert lib.gl_ctx_create(0, 1 << 20, ctypes.byref(ctx)) != 0
    assert b"HIP" in lib.gl_last_error() or b"device" in lib.gl_last_error()


@pytest.mark.parametrize("name,kind,inpaint", [("unet_small_text", "text", False), ("unet_small_text_image", "text_image", False),
                                               ("unet_small_keypoint", "keypoint", False), ("unet_small_inpaint", "text", True)])
def test_unet_state_dict_contract(name, kind, inpaint):
    """Same keys, shapes and order as the reference's UNetModel.state_dict()."""
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    m = UNetModel(**dict(syn.UNET_CFG_SMALL, grounding_tokenizer=syn.GROUNDING_TOKENIZERS[kind], inpaint_mode=inpaint))
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    ref = golden_shapes(name)
    assert list(mine) == sorted(mine, key=list(mine).index) and mine == ref
    assert m.in_channels == 4 and m.image_size == 64 and m.first_conv_restorable == (not inpaint)


def test_fuser_variants():
    """gatedCA (attention.py:190-212) and gatedSA2 (attention.py:251-297) keep the reference's keys and scale semantics."""
    from ldm.modules.attention import GatedCrossAttentionDense
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    cfg = dict(syn.UNET_CFG_SMALL, grounding_tokenizer=syn.GROUNDING_TOKENIZERS["text"])
    m = UNetModel(**dict(cfg, fuser_type="gatedCA"))
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == golden_shapes("unet_small_gatedca")
    assert any(type(x) is GatedCrossAttentionDense for x in m.modules())
    assert not any(k.endswith("fuser.linear.weight") for k in m.state_dict())
    m2 = UNetModel(**dict(cfg, fuser_type="gatedSA2"))
    assert {k: list(v.shape) for k, v in m2.state_dict().items()} == golden_shapes("unet_small_gatedsa2")
    # the reference's set_alpha_scale matches the other two classes by exact type and leaves GatedSelfAttentionDense2 alone
    from gligen_inference import set_alpha_scale
    set_alpha_scale(m2, 0.3)
    assert m2.fuser_scale() == 1.0
    set_alpha_scale(m, 0.3)
    assert m.fuser_scale() == 0.3


def test_full_model_contracts_on_meta_device():
    from ldm.models.autoencoder import AutoencoderKL
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    with torch.device("meta"):
        m = UNetModel(**dict(syn.UNET_CFG, grounding_tokenizer=syn.GROUNDING_TOKENIZERS["text"]))
        ae = AutoencoderKL(ddconfig=syn.VAE_DDCONFIG, embed_dim=4, scale_factor=0.18215)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == golden_shapes("unet_full_text")
    assert len(m.state_dict()) == 966
    assert {k: list(v.shape) for k, v in ae.state_dict().items()} == golden_shapes("vae_full")
    assert len(ae.state_dict()) == 248


def test_instantiate_from_config_and_yaml_targets():
    from ldm.util import get_obj_from_str, instantiate_from_config
    d = instantiate_from_config(dict(target="ldm.models.diffusion.ldm.LatentDiffusion", params=dict(linear_start=0.00085, linear_end=0.012, timesteps=1000)))
    m = load_golden("misc")
    for k, v in d.state_dict().items():
        np.testing.assert_allclose(v.numpy(), m["diff_" + k], rtol=1e-6, atol=1e-12)
    assert len(d.state_dict()) == 12
    for target in ["ldm.modules.diffusionmodules.openaimodel.UNetModel", "ldm.models.autoencoder.AutoencoderKL",
                   "ldm.modules.diffusionmodules.text_grounding_net.PositionNet", "ldm.modules.diffusionmodules.text_image_grounding_net.PositionNet",
                   "ldm.modules.diffusionmodules.keypoint_grounding_net.PositionNet", "ldm.models.diffusion.plms.PLMSSampler", *GINPUT.values()]:
        assert get_obj_from_str(target) is not None
    with pytest.raises(KeyError):
        instantiate_from_config({})
    assert instantiate_from_config("__is_first_stage__") is None


@pytest.mark.parametrize("kind", ["text", "text_image", "keypoint"])
def test_grounding_input_adapters(kind):
    from ldm.util import instantiate_from_config
    gi = instantiate_from_config(dict(target=GINPUT[kind]))
    assert gi.set is False
    with pytest.raises(AssertionError):
        gi.get_null_input()
    batch = syn.make_batch(kind, 3, n_valid=2)
    out = gi.prepare(batch)
    assert gi.set is True
    expect = {"text": {"boxes", "masks", "positive_embeddings"},
              "text_image": {"boxes", "masks", "text_masks", "image_masks", "text_embeddings", "image_embeddings"},
              "keypoint": {"points", "masks"}}[kind]
    assert set(out) == expect
    null = gi.get_null_input()
    assert set(null) == expect
    for k in expect:
        assert null[k].shape == out[k].shape and float(null[k].abs().sum()) == 0
    assert gi.get_null_input(batch=5)["masks"].shape[0] == 5
    if kind == "keypoint":
        assert gi.max_persons_per_image == 8


def test_alpha_generator_and_masks_match_reference():
    from gligen_inference import alpha_generator, draw_masks_from_boxes
    m = load_golden("misc")
    for S in (20, 50):
        for tp, tag in ((None, "none"), ([0.3, 0.0, 0.7], "0.3_0.0_0.7"), ([0.5, 0.25, 0.25], "0.5_0.25_0.25")):
            np.testing.assert_allclose(np.asarray(alpha_generator(S, tp), dtype=np.float64), m[f"alpha_{S}_{tag}"])
    with pytest.raises(AssertionError):
        alpha_generator(10, [0.5, 0.2, 0.2])
    assert np.array_equal(draw_masks_from_boxes(torch.from_numpy(m["mask_boxes"]), 64).numpy(), m["mask64"])


class _Recorder(torch.nn.Module):
    """Same mock as oracle/make_golden.py:Recorder, built on this repo's fuser class."""

    def __init__(self):
        super().__init__()
        from ldm.modules.attention import GatedSelfAttentionDense
        self.fuser = GatedSelfAttentionDense(8, 8, 1, 8)
        self.calls, self.restores = [], 0

    def restore_first_conv_from_SD(self):
        self.restores += 1

    def forward(self, inp):
        self.calls.append((int(inp["timesteps"][0]), "grounding_input" in inp, float(self.fuser.scale)))
        return torch.tanh(inp["x"]) * (0.5 if "grounding_input" in inp else 0.3) + 0.01 * inp["timesteps"].float().view(-1, 1, 1, 1) / 1000


@pytest.mark.parametrize("name", ["plms_trace_50", "plms_trace_20"])
def test_sampler_orchestration_matches_reference_trace(name):
    """PLMSSampler + set_alpha_scale + alpha_generator drive an arbitrary model exactly like the reference:
    same (timestep, cond/uncond, gate) sequence, same restore count, same schedule, same final latent."""
    from functools import partial
    from gligen_inference import alpha_generator, set_alpha_scale
    from ldm.models.diffusion.ldm import LatentDiffusion
    from ldm.models.diffusion.plms import PLMSSampler
    tr = load_golden(name)
    S, atype = tr["meta"]["S"], tr["meta"]["alpha_type"]
    mock = _Recorder()
    sampler = PLMSSampler(LatentDiffusion(linear_start=0.00085, linear_end=0.012, timesteps=1000), mock,
                          alpha_generator_func=partial(alpha_generator, type=atype), set_alpha_scale=set_alpha_scale)
    x = syn.make_latent(2, 4, 8, 8, seed=5)
    inp = dict(x=x.clone(), timesteps=None, context=torch.zeros(2, 1, 1), grounding_input={}, inpainting_extra_input=None, grounding_extra_input=None)
    out = sampler.sample(S=S, shape=(2, 4, 8, 8), input=inp, uc=torch.ones(2, 1, 1), guidance_scale=7.5)
    assert [(int(a), bool(b), float(c)) for a, b, c in tr["calls"]] == mock.calls
    assert mock.restores == int(tr["restores"])
    assert np.array_equal(sampler.ddim_timesteps, tr["ddim_timesteps"])
    np.testing.assert_allclose(np.asarray(sampler.ddim_alphas), tr["ddim_alphas"], rtol=1e-6)
    np.testing.assert_allclose(np.asarray(sampler.ddim_alphas_prev), tr["ddim_alphas_prev"], rtol=1e-6)
    assert mse(out, tr["x_out"]) < 1e-10
    assert inp["x"] is out and int(inp["timesteps"][0]) == int(tr["calls"][-1][0])


def test_ddim_sampler_orchestration_matches_reference_trace():
    """DDIMSampler (ldm.models.diffusion.ddim, eta 0) drives an arbitrary model exactly like the reference's."""
    from functools import partial
    from gligen_inference import alpha_generator, set_alpha_scale
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.ldm import LatentDiffusion
    tr = load_golden("ddim_trace_25")
    S, atype = tr["meta"]["S"], tr["meta"]["alpha_type"]
    mock = _Recorder()
    sampler = DDIMSampler(LatentDiffusion(linear_start=0.00085, linear_end=0.012, timesteps=1000), mock,
                          alpha_generator_func=partial(alpha_generator, type=atype), set_alpha_scale=set_alpha_scale)
    x = syn.make_latent(2, 4, 8, 8, seed=5)
    inp = dict(x=x.clone(), timesteps=None, context=torch.zeros(2, 1, 1), grounding_input={}, inpainting_extra_input=None, grounding_extra_input=None)
    out = sampler.sample(S=S, shape=(2, 4, 8, 8), input=inp, uc=torch.ones(2, 1, 1), guidance_scale=7.5)
    assert [(int(a), bool(b), float(c)) for a, b, c in tr["calls"]] == mock.calls
    assert mock.restores == int(tr["restores"])
    assert np.array_equal(sampler.ddim_timesteps, tr["ddim_timesteps"])
    assert mse(out, tr["x_out"]) < 1e-10
    assert inp["x"] is out and int(inp["timesteps"][0]) == int(tr["calls"][-1][0])
    with pytest.raises(NotImplementedError):
        sampler.make_schedule(10, ddim_eta=0.5)


def test_set_alpha_scale_is_exact_type_match():
    from gligen_inference import set_alpha_scale
    from ldm.modules.attention import GatedSelfAttentionDense

    class Sub(GatedSelfAttentionDense):
        pass

    net = torch.nn.ModuleList([GatedSelfAttentionDense(8, 8, 1, 8), Sub(8, 8, 1, 8)])
    set_alpha_scale(net, 0.25)
    assert net[0].scale == 0.25 and net[1].scale == 1


def test_prepare_batch_with_precomputed_features():
    import gligen_inference as gi
    gi_device, gi.device = gi.device, "cpu"
    try:
        meta = dict(locations=[[0.1, 0.2, 0.5, 0.6], [0.3, 0.3, 0.9, 0.8]], text_embeddings=[torch.ones(768), None], text_mask=[1, 0])
        b = gi.prepare_batch(meta, batch=3)
        assert b["boxes"].shape == (3, 30, 4) and b["masks"][0, :3].tolist() == [1, 1, 0]
        assert b["text_masks"][0, :3].tolist() == [1, 0, 0] and b["image_masks"].sum() == 0
        kp = gi.prepare_batch_kp(dict(locations=[[[0.5, 0.5]] * 17]), batch=2)
        assert kp["points"].shape == (2, 136, 2) and kp["masks"][0].sum() == 17
    finally:
        gi.device = gi_device
