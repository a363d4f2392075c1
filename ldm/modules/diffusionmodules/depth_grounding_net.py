"""depth grounding tokenizer (reference ldm/modules/diffusionmodules/depth_grounding_net.py:12-62): ConvNeXt-tiny tokens of the
depth map; forward kwargs (depth, mask)."""
from ldm.modules.diffusionmodules._spatial import SpatialPositionNet


class PositionNet(SpatialPositionNet):
    image_key = "depth"

    def __init__(self, resize_input=448, out_dim=768):
        super().__init__(resize_input=resize_input, out_dim=out_dim)
