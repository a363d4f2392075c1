"""depth downsampler (reference ldm/modules/diffusionmodules/depth_grounding_downsampler.py:9-29): bicubic resize to 256 x 256,
Conv2d(1, 4, 4, 2, 1) -> SiLU -> Conv2d(4, 8, 4, 2, 1)."""
from ldm.modules.diffusionmodules._spatial import SpatialDownsampler


class GroundingDownsampler(SpatialDownsampler):
    n_in = 1

    def __init__(self, resize_input=256, out_dim=8):
        super().__init__(resize_input=resize_input, out_dim=out_dim)
