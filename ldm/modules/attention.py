"""Transformer building blocks of the GLIGEN UNet — parameter containers.

Class names, constructor signatures and state_dict keys follow the reference's
ldm/modules/attention.py so checkpoints and `set_alpha_scale` keep working; the arithmetic
(attention.py:37-64 GEGLU, :127-149 cross-attention, :167-186 self-attention, :236-244 gated
self-attention, :333-338 block order, :366-376 SpatialTransformer) runs inside
UNetModel.forward as fused HIP kernels (gligen_amd/csrc/engine.hip: Engine::transformer).
"""
import torch
from torch import nn


class _EngineOnly(nn.Module):
    """Holds parameters; its math is executed by the native engine as part of UNetModel.forward."""

    def forward(self, *args, **kwargs):
        raise NotImplementedError(
            f"{type(self).__name__} is a parameter container: on MI355X its computation is fused into "
            "UNetModel.forward / AutoencoderKL.decode (libgligen_amd.so); call those instead")


def _slots(length, **at):
    """nn.Sequential of `length` entries with real modules at the given indices (keeps reference key numbering)."""
    mods = [nn.Identity() for _ in range(length)]
    for idx, m in at.items():
        mods[int(idx[1:])] = m
    return nn.Sequential(*mods)


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def Normalize(in_channels):
    return nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


class GEGLU(_EngineOnly):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)  # rows [0,dim_out) value, [dim_out,2*dim_out) gate


class FeedForward(_EngineOnly):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.0):
        super().__init__()
        if not glu:
            raise NotImplementedError("the GLIGEN UNet only uses the GEGLU feed-forward")
        inner = int(dim * mult)
        self.net = _slots(3, i0=GEGLU(dim, inner), i2=nn.Linear(inner, dim if dim_out is None else dim_out))


class CrossAttention(_EngineOnly):
    def __init__(self, query_dim, key_dim, value_dim, heads=8, dim_head=64, dropout=0):
        super().__init__()
        inner = dim_head * heads
        self.scale, self.heads = dim_head ** -0.5, heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(key_dim, inner, bias=False)
        self.to_v = nn.Linear(value_dim, inner, bias=False)
        self.to_out = _slots(2, i0=nn.Linear(inner, query_dim))


class SelfAttention(_EngineOnly):
    def __init__(self, query_dim, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = dim_head * heads
        self.scale, self.heads = dim_head ** -0.5, heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(query_dim, inner, bias=False)
        self.to_v = nn.Linear(query_dim, inner, bias=False)
        self.to_out = _slots(2, i0=nn.Linear(inner, query_dim))


class _GatedBase(_EngineOnly):
    def _gates(self, query_dim):
        self.ff = FeedForward(query_dim, glu=True)
        self.norm1 = nn.LayerNorm(query_dim)
        self.norm2 = nn.LayerNorm(query_dim)
        self.register_parameter("alpha_attn", nn.Parameter(torch.tensor(0.0)))
        self.register_parameter("alpha_dense", nn.Parameter(torch.tensor(0.0)))
        # external multiplier on tanh(alpha); set per sampling step by set_alpha_scale
        # (reference gligen_inference.py:24-28) and read by UNetModel.forward on every call
        self.scale = 1


class GatedCrossAttentionDense(_GatedBase):
    """fuser_type 'gatedCA' (reference attention.py:190-212): x += s*tanh(a1)*CA(LN(x), objs, objs); x += s*tanh(a2)*FF(LN(x)).
    No shipped GLIGEN config uses it; the engine runs it with the same kernels as attn2 (K / V of the grounding tokens
    are projected once per prompt)."""

    def __init__(self, query_dim, key_dim, value_dim, n_heads, d_head):
        super().__init__()
        self.attn = CrossAttention(query_dim=query_dim, key_dim=key_dim, value_dim=value_dim, heads=n_heads, dim_head=d_head)
        self._gates(query_dim)


class GatedSelfAttentionDense(_GatedBase):
    """The GLIGEN layer: x += s*tanh(a1)*SA(LN([x ; W objs]))[:, :N]; x += s*tanh(a2)*FF(LN(x))."""

    def __init__(self, query_dim, context_dim, n_heads, d_head):
        super().__init__()
        self.linear = nn.Linear(context_dim, query_dim)
        self.attn = SelfAttention(query_dim=query_dim, heads=n_heads, dim_head=d_head)
        self._gates(query_dim)


class GatedSelfAttentionDense2(_GatedBase):
    """fuser_type 'gatedSA2' (reference attention.py:251-297; the spatial-map modalities): same parameters as
    GatedSelfAttentionDense, but the residual is the attention output AT the grounding tokens (a square grid), resized
    bicubically to the visual grid. NB the reference's set_alpha_scale never touches this class (exact-type match on the
    other two, gligen_inference.py:24-28), so `scale` stays 1 unless set by hand."""

    def __init__(self, query_dim, context_dim, n_heads, d_head):
        super().__init__()
        self.linear = nn.Linear(context_dim, query_dim)
        self.attn = SelfAttention(query_dim=query_dim, heads=n_heads, dim_head=d_head)
        self._gates(query_dim)


class BasicTransformerBlock(_EngineOnly):
    def __init__(self, query_dim, key_dim, value_dim, n_heads, d_head, fuser_type, use_checkpoint=True):
        super().__init__()
        self.attn1 = SelfAttention(query_dim=query_dim, heads=n_heads, dim_head=d_head)
        self.ff = FeedForward(query_dim, glu=True)
        self.attn2 = CrossAttention(query_dim=query_dim, key_dim=key_dim, value_dim=value_dim, heads=n_heads, dim_head=d_head)
        self.norm1 = nn.LayerNorm(query_dim)
        self.norm2 = nn.LayerNorm(query_dim)
        self.norm3 = nn.LayerNorm(query_dim)
        self.use_checkpoint = use_checkpoint  # inert at inference, as in the reference
        if fuser_type == "gatedSA":
            self.fuser = GatedSelfAttentionDense(query_dim, key_dim, n_heads, d_head)
        elif fuser_type == "gatedCA":
            self.fuser = GatedCrossAttentionDense(query_dim, key_dim, value_dim, n_heads, d_head)
        elif fuser_type == "gatedSA2":
            self.fuser = GatedSelfAttentionDense2(query_dim, key_dim, n_heads, d_head)
        else:
            raise AssertionError(fuser_type)


class SpatialTransformer(_EngineOnly):
    def __init__(self, in_channels, key_dim, value_dim, n_heads, d_head, depth=1, fuser_type=None, use_checkpoint=True):
        super().__init__()
        if depth != 1:
            raise NotImplementedError("transformer_depth must be 1")
        self.in_channels = in_channels
        query_dim = n_heads * d_head
        self.norm = Normalize(in_channels)
        self.proj_in = nn.Conv2d(in_channels, query_dim, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(query_dim, key_dim, value_dim, n_heads, d_head, fuser_type, use_checkpoint=use_checkpoint)])
        self.proj_out = zero_module(nn.Conv2d(query_dim, in_channels, kernel_size=1, stride=1, padding=0))
