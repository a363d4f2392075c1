"""normal grounding tokenizer (reference ldm/modules/diffusionmodules/normal_grounding_net.py:12-62): ConvNeXt-tiny tokens of the
normal map; forward kwargs (normal, mask)."""
from ldm.modules.diffusionmodules._spatial import SpatialPositionNet


class PositionNet(SpatialPositionNet):
    image_key = "normal"

    def __init__(self, resize_input=448, out_dim=768):
        super().__init__(resize_input=resize_input, out_dim=out_dim)
