"""Keypoint grounding tokenizer: parameters of the reference PositionNet
(ldm/modules/diffusionmodules/keypoint_grounding_net.py:9-58): learned person + keypoint
embeddings, Fourier(xy), 8 persons x 17 COCO keypoints = 136 tokens. Executed in Engine::set_cond."""
import torch
import torch.nn as nn

from ldm.modules.attention import _EngineOnly
from ldm.modules.diffusionmodules.text_grounding_net import mlp
from ldm.modules.diffusionmodules.util import FourierEmbedder


class PositionNet(_EngineOnly):
    def __init__(self, max_persons_per_image, out_dim, fourier_freqs=8):
        super().__init__()
        if fourier_freqs != 8:
            raise NotImplementedError("fourier_freqs must be 8")
        self.max_persons_per_image, self.out_dim = max_persons_per_image, out_dim
        self.person_embeddings = nn.Parameter(torch.zeros([max_persons_per_image, out_dim]))
        self.keypoint_embeddings = nn.Parameter(torch.zeros([17, out_dim]))
        self.fourier_embedder = FourierEmbedder(num_freqs=fourier_freqs)
        self.position_dim = fourier_freqs * 2 * 2  # sin & cos of xy
        self.linears = mlp(out_dim + self.position_dim, out_dim)
        self.null_person_feature = nn.Parameter(torch.zeros([out_dim]))
        self.null_xy_feature = nn.Parameter(torch.zeros([self.position_dim]))
