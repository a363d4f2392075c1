"""AutoencoderKL (SD-1.4 KL-f8 VAE) — drop-in for the reference's ldm/models/autoencoder.py.

decode(z) = Decoder(post_quant_conv(z / scale_factor)) (autoencoder.py:40-44) runs entirely in the
native engine (Engine::vae_decode); encode(x) = posterior.sample() * scale_factor (autoencoder.py:34-38, the
inpainting configuration: once per prompt) runs in Engine::vae_encode.
"""
import torch
import torch.nn as nn

from gligen_amd import runtime as _rt
from ldm.modules.diffusionmodules.model import Decoder, Encoder


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, embed_dim, scale_factor=1):
        super().__init__()
        assert ddconfig["double_z"]
        self.ddconfig = dict(ddconfig)
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim
        self.scale_factor = scale_factor
        self._engine = None

    def _apply(self, fn, *a, **k):
        self._drop_engine()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._drop_engine()
        return super().load_state_dict(*a, **k)

    def _drop_engine(self):
        eng = self.__dict__.get("_engine")
        if eng is not None:
            eng.close()
        self.__dict__["_engine"] = None

    @property
    def engine(self):
        if self._engine is None:
            self._engine = _rt.build_vae_engine(self)
        return self._engine

    @torch.no_grad()
    def encode(self, x):
        """posterior.sample() * scale_factor (autoencoder.py:34-38). The reference draws the posterior noise with the CPU
        generator (distributions.py:35: torch.randn(mean.shape).to(device)); so does this, for seed-for-seed parity."""
        f = 2 ** (len(self.ddconfig["ch_mult"]) - 1)
        noise = torch.randn((x.shape[0], self.ddconfig["z_channels"], x.shape[2] // f, x.shape[3] // f))
        return self.engine.vae_encode(x, noise).to(x.dtype)

    @torch.no_grad()
    def decode(self, z):
        return self.engine.vae_decode(z).to(z.dtype)
