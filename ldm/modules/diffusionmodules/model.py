"""VAE encoder/decoder parameter containers (reference ldm/modules/diffusionmodules/model.py:
ResnetBlock 82-141, AttnBlock 150-202, Up/Downsample 42-79, Encoder 368-459, Decoder 462-568).

State_dict keys match the reference (248 tensors for the SD-1.4 KL-f8 autoencoder). The decoder
is executed by Engine::vae_decode (gligen_amd/csrc/engine.hip); nothing here runs torch ops.
"""
import numpy as np
import torch
import torch.nn as nn

from ldm.modules.attention import _EngineOnly


def Normalize(in_channels, num_groups=32):
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


def conv3(cin, cout, stride=1, padding=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=padding)


class Upsample(_EngineOnly):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = conv3(in_channels, in_channels)


class Downsample(_EngineOnly):
    """stride-2 conv after an asymmetric (0,1,0,1) zero pad (reference model.py:72-76)."""

    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = conv3(in_channels, in_channels, stride=2, padding=0)


class ResnetBlock(_EngineOnly):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels, self.use_conv_shortcut = in_channels, out_channels, conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = conv3(in_channels, out_channels)
        if temb_channels > 0:
            self.temb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = conv3(out_channels, out_channels)
        if in_channels != out_channels:
            if conv_shortcut:
                self.conv_shortcut = conv3(in_channels, out_channels)
            else:
                self.nin_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1)


class AttnBlock(_EngineOnly):
    """single-head attention over h*w positions with 1x1-conv projections."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        for name in ("q", "k", "v", "proj_out"):
            setattr(self, name, nn.Conv2d(in_channels, in_channels, kernel_size=1))


def make_attn(in_channels, attn_type="vanilla"):
    if attn_type == "none":
        return nn.Identity()
    if attn_type != "vanilla":
        raise NotImplementedError(f"attn_type {attn_type!r}")
    return AttnBlock(in_channels)


def _level(blocks, attn):
    lvl = nn.Module()
    lvl.block, lvl.attn = nn.ModuleList(blocks), nn.ModuleList(attn)
    return lvl


def _mid(ch, dropout, attn_type):
    mid = nn.Module()
    mid.block_1 = ResnetBlock(in_channels=ch, out_channels=ch, temb_channels=0, dropout=dropout)
    mid.attn_1 = make_attn(ch, attn_type=attn_type)
    mid.block_2 = ResnetBlock(in_channels=ch, out_channels=ch, temb_channels=0, dropout=dropout)
    return mid


class Encoder(_EngineOnly):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        self.ch, self.temb_ch, self.num_resolutions = ch, 0, len(ch_mult)
        self.num_res_blocks, self.resolution, self.in_channels = num_res_blocks, resolution, in_channels
        self.conv_in = conv3(in_channels, ch)
        self.in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        res, cin = resolution, ch
        for i, mult in enumerate(ch_mult):
            blocks, attn = [], []
            for _ in range(num_res_blocks):
                blocks.append(ResnetBlock(in_channels=cin, out_channels=ch * mult, temb_channels=0, dropout=dropout))
                cin = ch * mult
                if res in attn_resolutions:
                    attn.append(make_attn(cin, attn_type=attn_type))
            lvl = _level(blocks, attn)
            if i != self.num_resolutions - 1:
                lvl.downsample = Downsample(cin, resamp_with_conv)
                res //= 2
            self.down.append(lvl)
        self.mid = _mid(cin, dropout, attn_type)
        self.norm_out = Normalize(cin)
        self.conv_out = conv3(cin, 2 * z_channels if double_z else z_channels)


class Decoder(_EngineOnly):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        if give_pre_end or tanh_out:
            raise NotImplementedError("give_pre_end / tanh_out decoders are not used by GLIGEN")
        self.ch, self.temb_ch, self.num_resolutions = ch, 0, len(ch_mult)
        self.num_res_blocks, self.resolution, self.in_channels = num_res_blocks, resolution, in_channels
        cin = ch * ch_mult[-1]
        res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, res, res)
        self.conv_in = conv3(z_channels, cin)
        self.mid = _mid(cin, dropout, attn_type)
        ups = []
        for i in reversed(range(self.num_resolutions)):
            blocks, attn = [], []
            for _ in range(num_res_blocks + 1):
                blocks.append(ResnetBlock(in_channels=cin, out_channels=ch * ch_mult[i], temb_channels=0, dropout=dropout))
                cin = ch * ch_mult[i]
                if res in attn_resolutions:
                    attn.append(make_attn(cin, attn_type=attn_type))
            lvl = _level(blocks, attn)
            if i != 0:
                lvl.upsample = Upsample(cin, resamp_with_conv)
                res *= 2
            ups.insert(0, lvl)  # index = resolution level, 0 = full resolution
        self.up = nn.ModuleList(ups)
        self.norm_out = Normalize(cin)
        self.conv_out = conv3(cin, out_ch)
