"""Box + text + image grounding tokenizer: parameters of the reference PositionNet
(ldm/modules/diffusionmodules/text_image_grounding_net.py:9-65): two MLPs, 2N tokens
(text tokens first, then image tokens). Executed in Engine::set_cond."""
import torch
import torch.nn as nn

from ldm.modules.attention import _EngineOnly
from ldm.modules.diffusionmodules.text_grounding_net import mlp
from ldm.modules.diffusionmodules.util import FourierEmbedder


class PositionNet(_EngineOnly):
    def __init__(self, in_dim, out_dim, fourier_freqs=8):
        super().__init__()
        if fourier_freqs != 8:
            raise NotImplementedError("fourier_freqs must be 8")
        self.in_dim, self.out_dim = in_dim, out_dim
        self.fourier_embedder = FourierEmbedder(num_freqs=fourier_freqs)
        self.position_dim = fourier_freqs * 2 * 4
        self.linears_text = mlp(in_dim + self.position_dim, out_dim)
        self.linears_image = mlp(in_dim + self.position_dim, out_dim)
        self.null_text_feature = nn.Parameter(torch.zeros([in_dim]))
        self.null_image_feature = nn.Parameter(torch.zeros([in_dim]))
        self.null_position_feature = nn.Parameter(torch.zeros([self.position_dim]))
