"""Box+text grounding tokenizer: parameters of the reference PositionNet
(ldm/modules/diffusionmodules/text_grounding_net.py:9-47). Fourier(box) (+) phrase embedding,
null mixing and the 3-layer SiLU MLP run on the device in Engine::set_cond
(gligen_amd/csrc/misc.hip: posnet_input_kernel + gemm.hip)."""
import torch
import torch.nn as nn

from ldm.modules.attention import _EngineOnly, _slots
from ldm.modules.diffusionmodules.util import FourierEmbedder


def mlp(in_dim, out_dim, hidden=512):
    return _slots(5, i0=nn.Linear(in_dim, hidden), i2=nn.Linear(hidden, hidden), i4=nn.Linear(hidden, out_dim))


class PositionNet(_EngineOnly):
    def __init__(self, in_dim, out_dim, fourier_freqs=8):
        super().__init__()
        if fourier_freqs != 8:
            raise NotImplementedError("fourier_freqs must be 8")
        self.in_dim, self.out_dim = in_dim, out_dim
        self.fourier_embedder = FourierEmbedder(num_freqs=fourier_freqs)
        self.position_dim = fourier_freqs * 2 * 4  # sin & cos of xyxy
        self.linears = mlp(in_dim + self.position_dim, out_dim)
        self.null_positive_feature = nn.Parameter(torch.zeros([in_dim]))
        self.null_position_feature = nn.Parameter(torch.zeros([self.position_dim]))
