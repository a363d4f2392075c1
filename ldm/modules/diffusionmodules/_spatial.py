"""Shared pieces of the five spatial-map modalities (canny / hed / depth / normal / sem): their reference modules
differ only in the name of the conditioning image, the channels they read and a first 152 -> 3 conv for semantic maps
(reference ldm/modules/diffusionmodules/{canny,hed,depth,normal,sem}_grounding_{net,downsampler}.py)."""
import torch
import torch.nn as nn

from gligen_amd import runtime as _rt
from ldm.modules.attention import _EngineOnly, _slots
from ldm.modules.diffusionmodules.convnext import convnext_tiny


class SpatialPositionNet(_EngineOnly):
    """ConvNeXt-tiny features of the conditioning map (resized to resize_input, 1/32 resolution -> (resize_input/32)^2
    tokens of 768), null-feature mixing by `mask`, + learned position embedding, 3-layer SiLU MLP
    (reference canny_grounding_net.py:12-62). Step-invariant: evaluated once per prompt."""

    image_key = None   # name of the conditioning image in grounding_input
    in_dim = None      # sem: one-hot class planes, folded to 3 channels by in_conv first (sem_grounding_net.py:21,46)
    resize_mode = "nearest"

    def __init__(self, resize_input=448, out_dim=768, in_dim=None):
        super().__init__()
        self.resize_input, self.down_factor, self.out_dim = resize_input, 32, out_dim
        assert resize_input % self.down_factor == 0
        if in_dim is not None:
            self.in_dim = in_dim
            self.in_conv = nn.Conv2d(in_dim, 3, 3, 1, 1)
        self.convnext_tiny_backbone = convnext_tiny(pretrained=True)
        self.num_tokens = (resize_input // self.down_factor) ** 2
        feat = 768
        self.pos_embedding = nn.Parameter(torch.empty(1, self.num_tokens, feat).normal_(std=0.02))
        self.linears = _slots(5, i0=nn.Linear(feat, 512), i2=nn.Linear(512, 512), i4=nn.Linear(512, out_dim))
        self.null_feature = nn.Parameter(torch.zeros([feat]))

    def tokens(self, engine=None, **grounding_input):
        """PositionNet.forward(<image>, mask) -> grounding tokens [B, num_tokens, out_dim], computed by the engine that holds
        this model's weights (gl_op_spatial_tokens: ConvNeXt on the device). Precomputed tokens (`tokens=`) pass through."""
        if "tokens" in grounding_input:
            return grounding_input["tokens"]
        if engine is None:
            raise RuntimeError(f"{type(self).__module__}.PositionNet runs inside UNetModel on the MI355X engine (no CPU implementation)")
        return engine.spatial_tokens(grounding_input[self.image_key], grounding_input["mask"])


class SpatialDownsampler(nn.Module):
    """GroundingDownsampler (reference canny_grounding_downsampler.py:9-29 and siblings): the conditioning map, resized
    and run through two stride-2 4x4 convs, becomes `out_dim` extra input channels of the UNet's first conv
    (openaimodel.py:296-305, 442-444). forward() runs on the device through gl_op_grounding_downsample."""

    n_in = 1            # channels of grounding_extra_input that are read (grey maps are stored as RGB: channel 0)
    c_mid = 4
    mode = "bicubic"
    has_layers = True

    def __init__(self, resize_input=256, out_dim=8, in_dim=None):
        super().__init__()
        self.resize_input, self.out_dim = resize_input, out_dim
        if in_dim is not None:
            self.n_in = in_dim
        if self.has_layers:
            self.layers = nn.Sequential(nn.Conv2d(self.n_in, self.c_mid, 4, 2, 1), nn.SiLU(), nn.Conv2d(self.c_mid, out_dim, 4, 2, 1))
        self._eng = None

    @torch.no_grad()
    def forward(self, grounding_extra_input, engine=None):
        x = grounding_extra_input
        if not x.is_cuda:
            raise RuntimeError("GroundingDownsampler runs on the MI355X engine only (no CPU implementation)")
        eng = engine if engine is not None else _rt.scratch_engine(x.device)
        if self.has_layers:
            c1, c2 = self.layers[0], self.layers[2]
            return eng.grounding_downsample(x, self.n_in, self.resize_input, self.mode, (c1.weight, c1.bias, c2.weight, c2.bias))
        return eng.grounding_downsample(x, self.n_in, self.resize_input, self.mode, None)
