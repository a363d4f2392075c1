"""Semantic-map downsampler (reference ldm/modules/diffusionmodules/sem_grounding_downsampler.py:9-29): nearest resize of the
152 class planes to 256 x 256, Conv2d(152, 16, 4, 2, 1) -> SiLU -> Conv2d(16, 8, 4, 2, 1)."""
from ldm.modules.diffusionmodules._spatial import SpatialDownsampler


class GroundingDownsampler(SpatialDownsampler):
    c_mid = 16
    mode = "nearest"

    def __init__(self, resize_input=256, in_dim=152, out_dim=8):
        super().__init__(resize_input=resize_input, out_dim=out_dim, in_dim=in_dim)
