"""HED-edge downsampler (reference ldm/modules/diffusionmodules/hed_grounding_downsampler.py:9-22): no parameters, channel 0
of the edge map resized bicubically to 64 x 64."""
from ldm.modules.diffusionmodules._spatial import SpatialDownsampler


class GroundingDownsampler(SpatialDownsampler):
    has_layers = False

    def __init__(self, out_dim=1):
        super().__init__(resize_input=64, out_dim=out_dim)
