"""Glue between the reference-shaped nn.Module containers (ldm.*) and the native engine."""
from __future__ import annotations

from typing import Dict, Mapping, Optional

import torch

from .engine import Engine


def tensor_fingerprint(t: Optional[torch.Tensor]):
    if t is None:
        return None
    return (t.data_ptr(), t._version, tuple(t.shape), t.dtype, str(t.device))


def mapping_fingerprint(m: Mapping[str, torch.Tensor]):
    return tuple((k, tensor_fingerprint(v)) for k, v in sorted(m.items()))


def module_device(module: torch.nn.Module) -> torch.device:
    p = next(module.parameters())
    if not p.is_cuda:
        raise RuntimeError(
            f"{type(module).__name__} is on {p.device}: the GLIGEN MI355X path only runs on a HIP device "
            "(call .to('cuda') as gligen_inference.load_ckpt does); there is no CPU implementation")
    return p.device


def grounding_kind_of(position_net) -> str:
    mod = type(position_net).__module__.rsplit(".", 1)[-1]
    kinds = {"text_grounding_net": "text", "text_image_grounding_net": "text_image", "keypoint_grounding_net": "keypoint",
             # spatial-map tokenizers (ConvNeXt backbone, once per prompt): the engine takes their output tokens
             "canny_grounding_net": "tokens", "hed_grounding_net": "tokens", "depth_grounding_net": "tokens",
             "normal_grounding_net": "tokens", "sem_grounding_net": "tokens"}
    if mod not in kinds:
        raise NotImplementedError(f"grounding tokenizer {type(position_net).__module__} is not implemented on MI355X")
    return kinds[mod]


_SCRATCH = {}


def scratch_engine(device) -> Engine:
    """A weight-less engine per device for stand-alone operator calls (GroundingDownsampler outside a UNetModel)."""
    dev = torch.device(device)
    key = dev.index or 0
    if key not in _SCRATCH:
        _SCRATCH[key] = Engine(dev, arena_gb=1.0)
    return _SCRATCH[key]


def build_unet_engine(model, arena_gb: float = 12.0) -> Engine:
    dev = module_device(model)
    eng = Engine(dev, arena_gb=arena_gb)
    kind = grounding_kind_of(model.position_net)
    pn = model.position_net
    eng.configure_unet(
        in_channels=model.in_channels, out_channels=model.out_channels, model_channels=model.model_channels,
        num_res_blocks=model.num_res_blocks, num_heads=model.num_heads, context_dim=model.context_dim,
        channel_mult=list(model.channel_mult), attention_resolutions=list(model.attention_resolutions),
        inpaint_mode=model.inpaint_mode, grounding_kind=kind,
        gr_in_dim=getattr(pn, "in_dim", None) or pn.out_dim, gr_out_dim=pn.out_dim,
        max_persons=getattr(pn, "max_persons_per_image", 0), fuser_type=model.fuser_type,
        extra_channels=model.additional_channel_from_downsampler if model.first_conv_type == "GLIGEN" else 0,
        tok_resize=getattr(pn, "resize_input", 0) if kind == "tokens" else 0,
        tok_in_dim=(getattr(pn, "in_dim", None) or 0) if kind == "tokens" else 0)
    # (the GroundingDownsampler runs through its own operator call with its weights passed along: gl_op_grounding_downsample)
    eng.upload("unet", {k: v for k, v in model.state_dict().items() if not k.startswith("downsample_net.")})
    eng.finalize()
    return eng


def build_vae_engine(ae, arena_gb: float = 8.0) -> Engine:
    dev = module_device(ae)
    eng = Engine(dev, arena_gb=arena_gb)
    dd = ae.ddconfig
    eng.configure_vae(ch=dd["ch"], out_ch=dd["out_ch"], z_channels=dd["z_channels"], num_res_blocks=dd["num_res_blocks"],
                      embed_dim=ae.embed_dim, ch_mult=list(dd["ch_mult"]), scale_factor=ae.scale_factor)
    eng.upload("vae", ae.state_dict())  # decoder + post_quant_conv (decode), encoder + quant_conv (encode, inpainting)
    eng.finalize()
    return eng
