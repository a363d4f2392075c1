"""Deterministic synthetic weights and inputs (no checkpoints or datasets exist offline).

Weights are filled per key from a generator seeded by crc32(key) ^ seed, so the fill does not
depend on module construction order and the reference modules, the oracle and the HIP engine
all see identical tensors. Tensors the reference zero-initialises (zero_module convs, proj_out,
fuser gates, null features) get non-zero values — with real zeros eps would be identically 0
and the parity tests would be vacuous.
"""
from __future__ import annotations

import zlib
from typing import Dict, Mapping, Tuple

import torch

UNET_CFG = dict(image_size=64, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, transformer_depth=1, context_dim=768,
                fuser_type="gatedSA", use_checkpoint=True)
# a 2-level UNet with the same head dims (40, 80) for fast tests
UNET_CFG_SMALL = dict(UNET_CFG, channel_mult=[1, 2], attention_resolutions=[2, 1], num_res_blocks=1)
VAE_DDCONFIG = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
                    num_res_blocks=2, attn_resolutions=[], dropout=0.0)
VAE_DDCONFIG_SMALL = dict(VAE_DDCONFIG, ch=128, ch_mult=[1, 2], num_res_blocks=1)
GROUNDING_TOKENIZERS = {
    "text": dict(target="ldm.modules.diffusionmodules.text_grounding_net.PositionNet", params=dict(in_dim=768, out_dim=768)),
    "text_image": dict(target="ldm.modules.diffusionmodules.text_image_grounding_net.PositionNet", params=dict(in_dim=768, out_dim=768)),
    "keypoint": dict(target="ldm.modules.diffusionmodules.keypoint_grounding_net.PositionNet", params=dict(max_persons_per_image=8, out_dim=768)),
}


def _gen(key: str, seed: int) -> torch.Generator:
    return torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def _is_norm(key: str, shape) -> bool:
    if len(shape) != 1:
        return False
    name = ".".join(key.split(".")[-3:])
    return ("norm" in key.rsplit(".", 1)[0].rsplit(".", 1)[-1]) or name in (
        "in_layers.0.weight", "in_layers.0.bias", "out_layers.0.weight", "out_layers.0.bias", "out.0.weight", "out.0.bias")


def seeded_tensor(key: str, shape: Tuple[int, ...], seed: int = 1234, device="cpu") -> torch.Tensor:
    """The fixture value of one parameter. device='cpu' is the canonical fixture (what the goldens were made
    with); on a cuda device the same rules draw from the device generator (fast, for benchmarks)."""
    if str(device) == "cpu":
        g = _gen(key, seed)
    else:
        g = torch.Generator(device=device).manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    kw = dict(generator=g, device=device)
    shape = tuple(shape)
    if len(shape) == 0:  # fuser alpha_attn / alpha_dense: tanh(0.7) ~ 0.6 so the gated path matters
        return torch.tensor(0.7, device=device) + 0.2 * torch.randn((), **kw)
    if _is_norm(key, shape):
        if key.endswith(".weight"):
            return 1.0 + 0.1 * torch.randn(shape, **kw)
        return 0.05 * torch.randn(shape, **kw)
    if len(shape) >= 2 and "embeddings" not in key:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        bound = fan_in ** -0.5
        gain = 0.5 if any(t in key for t in ("out_layers.3.", "proj_out.", "out.2.")) else 1.0
        return (torch.rand(shape, **kw) * 2 - 1) * bound * gain * 1.7320508
    if "null_" in key or "embeddings" in key:
        return 0.5 * torch.randn(shape, **kw)
    return 0.02 * torch.randn(shape, **kw)  # biases


def fill_module_on_device_(module: torch.nn.Module, seed: int = 1234) -> torch.nn.Module:
    """Like fill_module_ but drawing on the module's (cuda) device: same distributions, not the same numbers."""
    with torch.no_grad():
        for k, v in module.state_dict().items():
            v.copy_(seeded_tensor(k, tuple(v.shape), seed, device=v.device))
    return module


def seeded_state_dict(shapes: Mapping[str, Tuple[int, ...]], seed: int = 1234) -> Dict[str, torch.Tensor]:
    return {k: seeded_tensor(k, tuple(s), seed) for k, s in shapes.items()}


def sd_first_conv_state(seed: int = 99) -> Dict[str, torch.Tensor]:
    """Seeded stand-in for the reference's SD_input_conv_weight_bias.pth (weight (320,4,3,3), bias (320,))."""
    return seeded_state_dict({"weight": (320, 4, 3, 3), "bias": (320,)}, seed)


def fill_module_(module: torch.nn.Module, seed: int = 1234) -> torch.nn.Module:
    """Overwrite every entry of module.state_dict() with its seeded value (buffers included)."""
    sd = module.state_dict()
    new = seeded_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed)
    module.load_state_dict(new, strict=True)
    return module


# ---- synthetic inputs ------------------------------------------------------
def make_boxes(B: int, n_valid: int, max_objs: int = 30, seed: int = 0):
    g = torch.Generator().manual_seed(1000 + seed)
    boxes = torch.zeros(B, max_objs, 4)
    masks = torch.zeros(B, max_objs)
    xy = torch.rand(B, n_valid, 2, generator=g) * 0.6
    wh = torch.rand(B, n_valid, 2, generator=g) * 0.3 + 0.1
    boxes[:, :n_valid, :2] = xy
    boxes[:, :n_valid, 2:] = (xy + wh).clamp(max=1.0)
    masks[:, :n_valid] = 1
    return boxes, masks


def make_embeddings(B: int, n_valid: int, max_objs: int = 30, dim: int = 768, seed: int = 0):
    g = torch.Generator().manual_seed(2000 + seed)
    e = torch.randn(B, max_objs, dim, generator=g)
    e = e / e.norm(dim=-1, keepdim=True) * 28.7  # CLIP feature norm used by the reference (gligen_inference.py:116)
    e[:, n_valid:] = 0
    return e


def make_batch(kind: str, B: int, n_valid: int = 8, seed: int = 0, max_objs: int = 30) -> Dict[str, torch.Tensor]:
    """A dataset-style batch dict (the input of GroundingNetInput.prepare)."""
    if kind == "keypoint":
        g = torch.Generator().manual_seed(3000 + seed)
        pts = torch.zeros(B, 8 * 17, 2)
        pts[:, : 2 * 17] = torch.rand(B, 2 * 17, 2, generator=g)
        return dict(points=pts, masks=(pts.mean(dim=2) != 0).float())
    boxes, masks = make_boxes(B, n_valid, max_objs=max_objs, seed=seed)
    batch = dict(boxes=boxes, masks=masks, text_embeddings=make_embeddings(B, n_valid, max_objs=max_objs, seed=seed))
    if kind == "text_image":
        batch.update(text_masks=masks.clone(), image_masks=masks.clone(),
                     image_embeddings=make_embeddings(B, n_valid, max_objs=max_objs, seed=seed + 7))
    return batch


def make_context(B: int, tokens: int = 77, dim: int = 768, seed: int = 0) -> torch.Tensor:
    return torch.randn(B, tokens, dim, generator=torch.Generator().manual_seed(4000 + seed))


def make_latent(B: int, C: int, h: int, w: int, seed: int = 0) -> torch.Tensor:
    return torch.randn(B, C, h, w, generator=torch.Generator().manual_seed(5000 + seed))


def make_spatial_map(modality: str, B: int, res: int, seed: int = 0) -> torch.Tensor:
    """A conditioning map as the reference's prepare_batch_* build it (gligen_inference.py:221-338): RGB-replicated grey
    map in [-1, 1] (canny / hed / depth), an RGB normal map, or 152 one-hot semantic class planes."""
    g = torch.Generator().manual_seed(6000 + seed)
    if modality == "sem":
        cls = torch.randint(0, 152, (B, res // 8, res // 8), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2)
        return torch.zeros(B, 152, res, res).scatter_(1, cls.unsqueeze(1), 1.0)
    if modality == "normal":
        return torch.rand(B, 3, res, res, generator=g) * 2 - 1
    return (torch.rand(B, 1, res, res, generator=g) * 2 - 1).repeat(1, 3, 1, 1)
