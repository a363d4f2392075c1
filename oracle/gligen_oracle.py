"""CPU fp32 oracle for the GLIGEN denoising path — TEST INFRASTRUCTURE ONLY.

A functional restatement (plain torch CPU ops over a state_dict, no nn.Module tree) of the
reference algorithm, each function citing the reference file:line it follows. Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(gligen_amd/, ldm/, grounding_input/, gligen_inference.py) never does.

Pinning: oracle/make_golden.py imports the real reference from /root/reference, runs it on the
seeded fixtures of gligen_amd.synthetic and stores its outputs under tests/golden/;
tests/test_oracle_golden.py checks this restatement against those files (no GPU needed). The
reference ships no tests or golden vectors of its own, so that execution is the only pin.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Mapping, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

SD = Mapping[str, torch.Tensor]


# --------------------------------------------------------------------------- small pieces
def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """[cos | sin] (cos first) — reference ldm/modules/diffusionmodules/util.py:160-180."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def fourier_embed(x: torch.Tensor, num_freqs: int = 8, temperature: float = 100.0) -> torch.Tensor:
    """[k][sin, cos][coord], f_k = temperature^(k/num_freqs) — reference util.py:12-26."""
    bands = temperature ** (torch.arange(num_freqs) / num_freqs)
    parts = []
    for f in bands:
        parts += [torch.sin(f * x), torch.cos(f * x)]
    return torch.cat(parts, dim=-1)


def _lin(sd: SD, p: str, x: torch.Tensor, bias: bool = True) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"] if bias else None)


def _mlp3(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    x = F.silu(_lin(sd, f"{p}.0", x))
    x = F.silu(_lin(sd, f"{p}.2", x))
    return _lin(sd, f"{p}.4", x)


def position_net(sd: SD, p: str, kind: str, g: Mapping[str, torch.Tensor]) -> torch.Tensor:
    """Grounding tokenizers: text_grounding_net.py:30-47, text_image_grounding_net.py:41-65,
    keypoint_grounding_net.py:34-58. `p` is the state_dict prefix ('position_net')."""
    if kind == "text":
        m = g["masks"].unsqueeze(-1)
        xy = fourier_embed(g["boxes"])
        feat = g["positive_embeddings"] * m + (1 - m) * sd[f"{p}.null_positive_feature"].view(1, 1, -1)
        xy = xy * m + (1 - m) * sd[f"{p}.null_position_feature"].view(1, 1, -1)
        return _mlp3(sd, f"{p}.linears", torch.cat([feat, xy], dim=-1))
    if kind == "text_image":
        m = g["masks"].unsqueeze(-1)
        tm, im = g["text_masks"].unsqueeze(-1), g["image_masks"].unsqueeze(-1)
        xy = fourier_embed(g["boxes"])
        te = g["text_embeddings"] * tm + (1 - tm) * sd[f"{p}.null_text_feature"].view(1, 1, -1)
        ie = g["image_embeddings"] * im + (1 - im) * sd[f"{p}.null_image_feature"].view(1, 1, -1)
        xy = xy * m + (1 - m) * sd[f"{p}.null_position_feature"].view(1, 1, -1)
        ot = _mlp3(sd, f"{p}.linears_text", torch.cat([te, xy], dim=-1))
        oi = _mlp3(sd, f"{p}.linears_image", torch.cat([ie, xy], dim=-1))
        return torch.cat([ot, oi], dim=1)
    if kind == "keypoint":
        m = g["masks"].unsqueeze(-1)
        pe, ke = sd[f"{p}.person_embeddings"], sd[f"{p}.keypoint_embeddings"]
        P = pe.shape[0]
        table = (pe.unsqueeze(1) + ke.unsqueeze(0)).reshape(P * 17, -1)  # person p, keypoint j -> row 17p+j
        feat = table.unsqueeze(0).expand(g["points"].shape[0], -1, -1)
        xy = fourier_embed(g["points"])
        feat = feat * m + (1 - m) * sd[f"{p}.null_person_feature"].view(1, 1, -1)
        xy = xy * m + (1 - m) * sd[f"{p}.null_xy_feature"].view(1, 1, -1)
        return _mlp3(sd, f"{p}.linears", torch.cat([feat, xy], dim=-1))
    raise ValueError(kind)


# --------------------------------------------------------------------------- spatial-map modalities
def _ln_channels_first(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """LayerNorm(data_format="channels_first") — reference convnext.py:141-146 (biased variance over C)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return w[:, None, None] * ((x - u) / torch.sqrt(s + eps)) + b[:, None, None]


def convnext_features(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """ConvNeXt.forward_features without head — reference convnext.py:108-112, Block.forward :36-50 (depths read off the keys)."""
    for i in range(4):
        d = f"{p}.downsample_layers.{i}"
        if i == 0:
            x = F.conv2d(x, sd[d + ".0.weight"], sd[d + ".0.bias"], stride=4)
            x = _ln_channels_first(x, sd[d + ".1.weight"], sd[d + ".1.bias"])
        else:
            x = _ln_channels_first(x, sd[d + ".0.weight"], sd[d + ".0.bias"])
            x = F.conv2d(x, sd[d + ".1.weight"], sd[d + ".1.bias"], stride=2)
        j = 0
        while f"{p}.stages.{i}.{j}.dwconv.weight" in sd:
            b = f"{p}.stages.{i}.{j}"
            C = x.shape[1]
            h = F.conv2d(x, sd[b + ".dwconv.weight"], sd[b + ".dwconv.bias"], padding=3, groups=C).permute(0, 2, 3, 1)
            h = F.layer_norm(h, (C,), sd[b + ".norm.weight"], sd[b + ".norm.bias"], 1e-6)
            h = F.linear(F.gelu(F.linear(h, sd[b + ".pwconv1.weight"], sd[b + ".pwconv1.bias"])), sd[b + ".pwconv2.weight"], sd[b + ".pwconv2.bias"])
            if b + ".gamma" in sd:
                h = sd[b + ".gamma"] * h
            x = x + h.permute(0, 3, 1, 2)
            j += 1
    return x


def spatial_position_net(sd: SD, p: str, image: torch.Tensor, mask: torch.Tensor, resize_input: int) -> torch.Tensor:
    """PositionNet.forward of the canny / hed / depth / normal / sem tokenizers — reference canny_grounding_net.py:38-62
    (sem_grounding_net.py:40-65 adds the nearest resize + in_conv in front)."""
    B = image.shape[0]
    x = F.interpolate(image, resize_input)  # default mode 'nearest' (sem passes it explicitly)
    if p + ".in_conv.weight" in sd:
        x = F.conv2d(x, sd[p + ".in_conv.weight"], sd[p + ".in_conv.bias"], padding=1)
    feat = convnext_features(sd, p + ".convnext_tiny_backbone", x)
    T = feat.shape[2] * feat.shape[3]
    objs = feat.reshape(B, -1, T).permute(0, 2, 1)
    m = mask.view(-1, 1, 1)
    objs = objs * m + sd[p + ".null_feature"].view(1, 1, -1) * (1 - m) + sd[p + ".pos_embedding"]
    return _mlp3(sd, p + ".linears", objs)


def grounding_downsampler(sd: SD, p: str, x: torch.Tensor, n_in: int, resize: int, mode: str = "bicubic") -> torch.Tensor:
    """GroundingDownsampler.forward — reference canny_grounding_downsampler.py:23-29 (hed: no layers, resize 64, :16-22;
    sem: all planes, nearest, sem_grounding_downsampler.py:23-28)."""
    out = F.interpolate(x[:, :n_in], (resize, resize), mode=mode)
    if p + ".layers.0.weight" in sd:
        out = F.conv2d(out, sd[p + ".layers.0.weight"], sd[p + ".layers.0.bias"], stride=2, padding=1)
        out = F.conv2d(F.silu(out), sd[p + ".layers.2.weight"], sd[p + ".layers.2.bias"], stride=2, padding=1)
    return out


def null_grounding(kind: str, g: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """get_null_input(): all-zero tensors of the prepared shapes —
    grounding_input/text_grounding_tokinzer_input.py:29-45 (and the text_image / keypoint twins)."""
    return {k: torch.zeros_like(v) for k, v in g.items()}


def _gn(sd: SD, p: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _mha(q, k, v, heads: int) -> torch.Tensor:
    """softmax(q k^T * d^-0.5) v with scale applied after the product — attention.py:134-147, 174-184."""
    B, N, C = q.shape
    d = C // heads
    q = q.view(B, N, heads, d).transpose(1, 2)
    k = k.view(B, -1, heads, d).transpose(1, 2)
    v = v.view(B, -1, heads, d).transpose(1, 2)
    sim = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    out = torch.matmul(sim.softmax(dim=-1), v)
    return out.transpose(1, 2).reshape(B, N, C)


def self_attention(sd: SD, p: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """SelfAttention.forward — attention.py:167-186 (q/k/v bias-free, to_out.0 with bias)."""
    o = _mha(_lin(sd, p + ".to_q", x, False), _lin(sd, p + ".to_k", x, False), _lin(sd, p + ".to_v", x, False), heads)
    return _lin(sd, p + ".to_out.0", o)


def cross_attention(sd: SD, p: str, x: torch.Tensor, ctx: torch.Tensor, heads: int) -> torch.Tensor:
    """CrossAttention.forward (mask=None) — attention.py:127-149."""
    o = _mha(_lin(sd, p + ".to_q", x, False), _lin(sd, p + ".to_k", ctx, False), _lin(sd, p + ".to_v", ctx, False), heads)
    return _lin(sd, p + ".to_out.0", o)


def feed_forward(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """GEGLU FeedForward: first half value, second half gate, exact-erf GELU — attention.py:37-64."""
    val, gate = _lin(sd, p + ".net.0.proj", x).chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", val * F.gelu(gate))


def _ln(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def gated_self_attention(sd: SD, p: str, x: torch.Tensor, objs: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """GatedSelfAttentionDense.forward — attention.py:236-244."""
    n = x.shape[1]
    o = _lin(sd, p + ".linear", objs)
    a = self_attention(sd, p + ".attn", _ln(sd, p + ".norm1", torch.cat([x, o], dim=1)), heads)[:, :n]
    x = x + scale * torch.tanh(sd[p + ".alpha_attn"]) * a
    x = x + scale * torch.tanh(sd[p + ".alpha_dense"]) * feed_forward(sd, p + ".ff", _ln(sd, p + ".norm2", x))
    return x


def gated_self_attention2(sd: SD, p: str, x: torch.Tensor, objs: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """GatedSelfAttentionDense2.forward (fuser_type 'gatedSA2') — attention.py:271-297: the attention outputs AT the
    grounding tokens (a size_g x size_g grid) are resized bicubically to the visual grid and added as the residual."""
    B, n, _ = x.shape
    ng = objs.shape[1]
    sv, sg = int(round(n ** 0.5)), int(round(ng ** 0.5))
    assert sv * sv == n and sg * sg == ng, "visual / grounding tokens must be square rootable (attention.py:279-283)"
    o = _lin(sd, p + ".linear", objs)
    out = self_attention(sd, p + ".attn", _ln(sd, p + ".norm1", torch.cat([x, o], dim=1)), heads)[:, n:]
    out = out.permute(0, 2, 1).reshape(B, -1, sg, sg)
    out = F.interpolate(out, (sv, sv), mode="bicubic")
    residual = out.reshape(B, -1, n).permute(0, 2, 1)
    x = x + scale * torch.tanh(sd[p + ".alpha_attn"]) * residual
    x = x + scale * torch.tanh(sd[p + ".alpha_dense"]) * feed_forward(sd, p + ".ff", _ln(sd, p + ".norm2", x))
    return x


def gated_cross_attention(sd: SD, p: str, x: torch.Tensor, objs: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """GatedCrossAttentionDense.forward (fuser_type 'gatedCA') — attention.py:207-212."""
    a = cross_attention(sd, p + ".attn", _ln(sd, p + ".norm1", x), objs, heads)
    x = x + scale * torch.tanh(sd[p + ".alpha_attn"]) * a
    x = x + scale * torch.tanh(sd[p + ".alpha_dense"]) * feed_forward(sd, p + ".ff", _ln(sd, p + ".norm2", x))
    return x


def transformer_block(sd: SD, p: str, x, ctx, objs, heads: int, scale: float, fuser_type: Optional[str] = None) -> torch.Tensor:
    """BasicTransformerBlock._forward — attention.py:333-338. The fuser variant is read off the state_dict: gatedSA has
    fuser.linear (attention.py:219), gatedCA does not (attention.py:190-205); gatedSA2 shares gatedSA's keys and is selected by `fuser_type`."""
    x = self_attention(sd, p + ".attn1", _ln(sd, p + ".norm1", x), heads) + x
    if fuser_type == "gatedSA2":
        x = gated_self_attention2(sd, p + ".fuser", x, objs, heads, scale)
    elif p + ".fuser.linear.weight" in sd:
        x = gated_self_attention(sd, p + ".fuser", x, objs, heads, scale)
    else:
        x = gated_cross_attention(sd, p + ".fuser", x, objs, heads, scale)
    x = cross_attention(sd, p + ".attn2", _ln(sd, p + ".norm2", x), ctx, heads) + x
    x = feed_forward(sd, p + ".ff", _ln(sd, p + ".norm3", x)) + x
    return x


def spatial_transformer(sd: SD, p: str, x, ctx, objs, heads: int, scale: float, fuser_type: Optional[str] = None) -> torch.Tensor:
    """SpatialTransformer.forward (GroupNorm eps 1e-6, 1x1 in/out projections) — attention.py:366-376."""
    B, C, H, W = x.shape
    h = _gn(sd, p + ".norm", x, 1e-6)
    h = F.conv2d(h, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    h = h.flatten(2).transpose(1, 2)
    h = transformer_block(sd, p + ".transformer_blocks.0", h, ctx, objs, heads, scale, fuser_type)
    h = h.transpose(1, 2).reshape(B, C, H, W)
    h = F.conv2d(h, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return h + x


def unet_resblock(sd: SD, p: str, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    """ResBlock._forward (GroupNorm32 eps 1e-5, no scale-shift, no up/down) — openaimodel.py:212-232."""
    h = F.conv2d(F.silu(_gn(sd, p + ".in_layers.0", x, 1e-5)), sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    h = h + _lin(sd, p + ".emb_layers.1", F.silu(emb))[:, :, None, None]
    h = F.conv2d(F.silu(_gn(sd, p + ".out_layers.0", h, 1e-5)), sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if p + ".skip_connection.weight" in sd:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


# --------------------------------------------------------------------------- UNet
def unet_forward(sd: SD, cfg: Mapping, inp: Mapping, fuser_scale: float = 1.0) -> torch.Tensor:
    """UNetModel.forward — openaimodel.py:420-464; topology as built in openaimodel.py:299-395.

    cfg: model_channels, channel_mult, num_res_blocks, attention_resolutions, num_heads, grounding_kind.
    inp: x, timesteps, context, grounding_input (PositionNet kwargs) and optionally inpainting_extra_input.
    A missing grounding_input is the caller's job (pass null_grounding(...)), as in openaimodel.py:422-426.
    """
    mc, heads = cfg["model_channels"], cfg["num_heads"]
    if not isinstance(fuser_scale, (int, float)):   # one `scale` per fuser module, in module order (they are plain attributes)
        scales = iter(list(fuser_scale))
    else:
        scales = None
    _scale = (lambda: next(scales)) if scales is not None else (lambda: fuser_scale)
    if cfg.get("grounding_kind") == "spatial":  # canny / hed / depth / normal / sem: {"image", "mask"} (or precomputed "tokens")
        gi = inp["grounding_input"]
        objs = gi["tokens"] if "tokens" in gi else spatial_position_net(sd, "position_net", gi["image"], gi["mask"], cfg["tok_resize"])
    else:
        objs = position_net(sd, "position_net", cfg.get("grounding_kind", "text"), inp["grounding_input"])
    emb = _lin(sd, "time_embed.2", F.silu(_lin(sd, "time_embed.0", timestep_embedding(inp["timesteps"], mc))))
    h = inp["x"]
    if inp.get("grounding_extra_input") is not None:  # downsampled conditioning map in front of the first conv — openaimodel.py:442-444
        ds_cfg = cfg["downsampler"]
        h = torch.cat([h, grounding_downsampler(sd, "downsample_net", inp["grounding_extra_input"], ds_cfg["n_in"], ds_cfg["resize"], ds_cfg["mode"])], dim=1)
    if inp.get("inpainting_extra_input") is not None:
        h = torch.cat([h, inp["inpainting_extra_input"]], dim=1)
    ctx = inp["context"]

    def attn_here(ds):
        return ds in cfg["attention_resolutions"]

    hs: List[torch.Tensor] = []
    h = F.conv2d(h, sd["input_blocks.0.0.weight"], sd["input_blocks.0.0.bias"], padding=1)
    hs.append(h)
    n, ds = 1, 1
    mults = list(cfg["channel_mult"])
    for level, _ in enumerate(mults):
        for _ in range(cfg["num_res_blocks"]):
            h = unet_resblock(sd, f"input_blocks.{n}.0", h, emb)
            if attn_here(ds):
                h = spatial_transformer(sd, f"input_blocks.{n}.1", h, ctx, objs, heads, _scale(), cfg.get("fuser_type"))
            hs.append(h)
            n += 1
        if level != len(mults) - 1:
            h = F.conv2d(h, sd[f"input_blocks.{n}.0.op.weight"], sd[f"input_blocks.{n}.0.op.bias"], stride=2, padding=1)
            hs.append(h)
            n += 1
            ds *= 2
    h = unet_resblock(sd, "middle_block.0", h, emb)
    h = spatial_transformer(sd, "middle_block.1", h, ctx, objs, heads, _scale(), cfg.get("fuser_type"))
    h = unet_resblock(sd, "middle_block.2", h, emb)
    n = 0
    for level in reversed(range(len(mults))):
        for i in range(cfg["num_res_blocks"] + 1):
            h = torch.cat([h, hs.pop()], dim=1)  # decoder activations first — openaimodel.py:461
            h = unet_resblock(sd, f"output_blocks.{n}.0", h, emb)
            j = 1
            if attn_here(ds):
                h = spatial_transformer(sd, f"output_blocks.{n}.1", h, ctx, objs, heads, _scale(), cfg.get("fuser_type"))
                j = 2
            if level and i == cfg["num_res_blocks"]:
                h = F.interpolate(h, scale_factor=2, mode="nearest")
                h = F.conv2d(h, sd[f"output_blocks.{n}.{j}.conv.weight"], sd[f"output_blocks.{n}.{j}.conv.bias"], padding=1)
                ds //= 2
            n += 1
    h = F.silu(_gn(sd, "out.0", h, 1e-5))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


# --------------------------------------------------------------------------- VAE
def vae_resblock(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """ResnetBlock.forward with temb=None (GroupNorm eps 1e-6, swish) — model.py:118-141."""
    h = F.conv2d(F.silu(_gn(sd, p + ".norm1", x, 1e-6)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(sd, p + ".norm2", h, 1e-6)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if p + ".nin_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + ".nin_shortcut.weight"], sd[p + ".nin_shortcut.bias"])
    return x + h


def vae_attn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """AttnBlock.forward: single head, scale C^-0.5 — model.py:177-202."""
    B, C, H, W = x.shape
    h = _gn(sd, p + ".norm", x, 1e-6)
    q = F.conv2d(h, sd[p + ".q.weight"], sd[p + ".q.bias"]).flatten(2).transpose(1, 2)
    k = F.conv2d(h, sd[p + ".k.weight"], sd[p + ".k.bias"]).flatten(2)
    v = F.conv2d(h, sd[p + ".v.weight"], sd[p + ".v.bias"]).flatten(2)
    w = torch.bmm(q, k) * (int(C) ** -0.5)
    w = F.softmax(w, dim=2)
    h = torch.bmm(v, w.transpose(1, 2)).reshape(B, C, H, W)
    return x + F.conv2d(h, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])


def vae_decode(sd: SD, cfg: Mapping, z: torch.Tensor) -> torch.Tensor:
    """AutoencoderKL.decode -> Decoder.forward — autoencoder.py:40-44, model.py:535-568."""
    z = z / cfg["scale_factor"]
    z = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    d = "decoder"
    h = F.conv2d(z, sd[f"{d}.conv_in.weight"], sd[f"{d}.conv_in.bias"], padding=1)
    h = vae_resblock(sd, f"{d}.mid.block_1", h)
    h = vae_attn(sd, f"{d}.mid.attn_1", h)
    h = vae_resblock(sd, f"{d}.mid.block_2", h)
    nres = len(cfg["ch_mult"])
    for level in reversed(range(nres)):
        for i in range(cfg["num_res_blocks"] + 1):
            h = vae_resblock(sd, f"{d}.up.{level}.block.{i}", h)
        if level != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"{d}.up.{level}.upsample.conv.weight"], sd[f"{d}.up.{level}.upsample.conv.bias"], padding=1)
    h = F.silu(_gn(sd, f"{d}.norm_out", h, 1e-6))
    return F.conv2d(h, sd[f"{d}.conv_out.weight"], sd[f"{d}.conv_out.bias"], padding=1)


def vae_encode(sd: SD, cfg: Mapping, x: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """AutoencoderKL.encode -> Encoder.forward -> quant_conv -> DiagonalGaussianDistribution.sample() * scale_factor —
    autoencoder.py:34-38, model.py:434-459 (Downsample: zero pad (0,1,0,1) then 3x3 stride-2 conv, model.py:72-76),
    distributions.py:24-37. `noise` stands for the reference's torch.randn(mean.shape) draw (CPU generator)."""
    e = "encoder"
    h = F.conv2d(x, sd[f"{e}.conv_in.weight"], sd[f"{e}.conv_in.bias"], padding=1)
    nres = len(cfg["ch_mult"])
    for level in range(nres):
        for i in range(cfg["num_res_blocks"]):
            h = vae_resblock(sd, f"{e}.down.{level}.block.{i}", h)
        if level != nres - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = F.conv2d(h, sd[f"{e}.down.{level}.downsample.conv.weight"], sd[f"{e}.down.{level}.downsample.conv.bias"], stride=2)
    h = vae_resblock(sd, f"{e}.mid.block_1", h)
    h = vae_attn(sd, f"{e}.mid.attn_1", h)
    h = vae_resblock(sd, f"{e}.mid.block_2", h)
    h = F.silu(_gn(sd, f"{e}.norm_out", h, 1e-6))
    h = F.conv2d(h, sd[f"{e}.conv_out.weight"], sd[f"{e}.conv_out.bias"], padding=1)
    moments = F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])
    mean, logvar = torch.chunk(moments, 2, dim=1)
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
    return (mean + std * noise) * cfg["scale_factor"]


def to_uint8(img: torch.Tensor) -> np.ndarray:
    """clamp(-1,1)*0.5+0.5, *255, astype(uint8) truncation, HWC — gligen_inference.py:443-445."""
    s = torch.clamp(img, min=-1, max=1) * 0.5 + 0.5
    return (s.cpu().numpy().transpose(0, 2, 3, 1) * 255).astype(np.uint8)


# --------------------------------------------------------------------------- schedule + sampler
def make_schedule(linear_start: float = 0.00085, linear_end: float = 0.012, timesteps: int = 1000) -> Dict[str, np.ndarray]:
    """'linear' beta schedule in fp64, cumprod, fp32 buffers — util.py:30-34, ddpm.py:19-45."""
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64) ** 2).numpy()
    ac = np.cumprod(1.0 - betas, axis=0)
    return {
        "betas": betas.astype(np.float32),
        "alphas_cumprod": ac.astype(np.float32),
        "sqrt_alphas_cumprod": np.sqrt(ac).astype(np.float32),
        "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac).astype(np.float32),
    }


def plms_schedule(S: int, sched: Mapping[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """ddim_timesteps = arange(0, T, T//S) + 1; alphas, alphas_prev — util.py:55-83, plms.py:25-56."""
    T = len(sched["alphas_cumprod"])
    ts = np.asarray(list(range(0, T, T // S))) + 1
    ac = sched["alphas_cumprod"]
    return {"ddim_timesteps": ts, "ddim_alphas": ac[ts], "ddim_alphas_prev": np.asarray([ac[0]] + ac[ts[:-1]].tolist(), dtype=np.float32)}


def alpha_generator(length: int, type: Optional[Sequence[float]] = None) -> List[float]:
    """Gate schedule [1]*s0 + linear decay + [0]*s2 — gligen_inference.py:31-66."""
    if type is None:
        type = [1, 0, 0]
    assert len(type) == 3 and type[0] + type[1] + type[2] == 1
    s0, s1 = int(type[0] * length), int(type[1] * length)
    s2 = length - s0 - s1
    decay = list(np.arange(start=0, stop=1, step=1 / s1)[::-1]) if s1 != 0 else []
    out = [1] * s0 + decay + [0] * s2
    assert len(out) == length
    return out


def plms_sample(eps_fn: Callable[[torch.Tensor, torch.Tensor, bool, float], torch.Tensor], x: torch.Tensor, S: int,
                sched: Mapping[str, np.ndarray], guidance_scale: float, alphas: Optional[Sequence[float]] = None,
                mask: Optional[torch.Tensor] = None, x0: Optional[torch.Tensor] = None,
                noise: Optional[torch.Tensor] = None, on_gate_off: Optional[Callable[[], None]] = None) -> torch.Tensor:
    """PLMSSampler.plms_sampling + p_sample_plms (sigma = 0) — plms.py:65-162.

    eps_fn(x, t, cond, fuser_scale) evaluates the UNet with (cond=True) or without grounding/text
    conditioning. `noise[i]` replaces the q_sample draw of step i when inpainting (plms.py:96-100).
    `on_gate_off()` stands for model.restore_first_conv_from_SD(), called at every step whose gate
    scale is 0 (plms.py:88-89).
    """
    ps = plms_schedule(S, sched)
    time_range = np.flip(ps["ddim_timesteps"])
    b = x.shape[0]
    old_eps: List[torch.Tensor] = []

    def model_out(xc, t, scale):
        e = eps_fn(xc, t, True, scale)
        if guidance_scale != 1:
            eu = eps_fn(xc, t, False, scale)
            e = eu + guidance_scale * (e - eu)
        return e

    def x_prev_of(xc, e, index):
        a_t, a_prev = float(ps["ddim_alphas"][index]), float(ps["ddim_alphas_prev"][index])
        pred_x0 = (xc - math.sqrt(1.0 - a_t) * e) / math.sqrt(a_t)
        return math.sqrt(a_prev) * pred_x0 + math.sqrt(1.0 - a_prev) * e

    for i, step in enumerate(time_range):
        scale = 1.0 if alphas is None else float(alphas[i])
        if alphas is not None and alphas[i] == 0 and on_gate_off is not None:
            on_gate_off()
        index = len(time_range) - i - 1
        ts = torch.full((b,), int(step), dtype=torch.long)
        ts_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), dtype=torch.long)
        if mask is not None:
            sa = float(sched["sqrt_alphas_cumprod"][int(step)])
            s1 = float(sched["sqrt_one_minus_alphas_cumprod"][int(step)])
            x = (sa * x0 + s1 * noise[i]) * mask + (1.0 - mask) * x
        e_t = model_out(x, ts, scale)
        if len(old_eps) == 0:
            e_next = model_out(x_prev_of(x, e_t, index), ts_next, scale)
            e_prime = (e_t + e_next) / 2
        elif len(old_eps) == 1:
            e_prime = (3 * e_t - old_eps[-1]) / 2
        elif len(old_eps) == 2:
            e_prime = (23 * e_t - 16 * old_eps[-1] + 5 * old_eps[-2]) / 12
        else:
            e_prime = (55 * e_t - 59 * old_eps[-1] + 37 * old_eps[-2] - 9 * old_eps[-3]) / 24
        x = x_prev_of(x, e_prime, index)
        old_eps.append(e_t)
        if len(old_eps) >= 4:
            old_eps.pop(0)
    return x


def ddim_sample(eps_fn: Callable[[torch.Tensor, torch.Tensor, bool, float], torch.Tensor], x: torch.Tensor, S: int,
                sched: Mapping[str, np.ndarray], guidance_scale: float, alphas: Optional[Sequence[float]] = None,
                mask: Optional[torch.Tensor] = None, x0: Optional[torch.Tensor] = None,
                noise: Optional[torch.Tensor] = None, on_gate_off: Optional[Callable[[], None]] = None) -> torch.Tensor:
    """DDIMSampler.ddim_sampling + p_sample_ddim with eta = 0 (the only value the reference uses: make_schedule's
    default, ddim.py:27,59-62) — ddim.py:65-134. One model evaluation pair per step, no multistep history:
    x_prev = sqrt(a_prev) * pred_x0 + sqrt(1 - a_prev) * e_t (sigma_t = 0, so the randn term vanishes)."""
    ps = plms_schedule(S, sched)  # same ddim_timesteps / ddim_alphas / ddim_alphas_prev as the PLMS sampler (util.py:55-83)
    time_range = np.flip(ps["ddim_timesteps"])
    b = x.shape[0]
    for i, step in enumerate(time_range):
        scale = 1.0 if alphas is None else float(alphas[i])
        if alphas is not None and alphas[i] == 0 and on_gate_off is not None:
            on_gate_off()
        index = len(time_range) - i - 1
        ts = torch.full((b,), int(step), dtype=torch.long)
        if mask is not None:
            sa = float(sched["sqrt_alphas_cumprod"][int(step)])
            s1 = float(sched["sqrt_one_minus_alphas_cumprod"][int(step)])
            x = (sa * x0 + s1 * noise[i]) * mask + (1.0 - mask) * x
        e = eps_fn(x, ts, True, scale)
        if guidance_scale != 1:
            eu = eps_fn(x, ts, False, scale)
            e = eu + guidance_scale * (e - eu)
        a_t, a_prev = float(ps["ddim_alphas"][index]), float(ps["ddim_alphas_prev"][index])
        pred_x0 = (x - math.sqrt(1.0 - a_t) * e) / math.sqrt(a_t)
        x = math.sqrt(a_prev) * pred_x0 + math.sqrt(1.0 - a_prev) * e
    return x


def draw_masks_from_boxes(boxes: torch.Tensor, size: int) -> torch.Tensor:
    """1 everywhere, 0 inside int(box*size) rectangles — inpaint_mask_func.py:16-41."""
    out = []
    for bx in boxes:
        m = torch.ones(size, size)
        for box in bx:
            x0, y0, x1, y1 = box * size
            m[int(y0):int(y1), int(x0):int(x1)] = 0
        out.append(m)
    return torch.stack(out).unsqueeze(1)
