"""hed grounding tokenizer (reference ldm/modules/diffusionmodules/hed_grounding_net.py:12-62): ConvNeXt-tiny tokens of the
hed map; forward kwargs (hed_edge, mask)."""
from ldm.modules.diffusionmodules._spatial import SpatialPositionNet


class PositionNet(SpatialPositionNet):
    image_key = "hed_edge"

    def __init__(self, resize_input=448, out_dim=768):
        super().__init__(resize_input=resize_input, out_dim=out_dim)
