"""Semantic-map grounding tokenizer (reference ldm/modules/diffusionmodules/sem_grounding_net.py:12-65): one-hot class
planes (152) -> nearest resize -> in_conv 152 -> 3 -> ConvNeXt-tiny tokens."""
from ldm.modules.diffusionmodules._spatial import SpatialPositionNet


class PositionNet(SpatialPositionNet):
    image_key = "sem"

    def __init__(self, resize_input=448, in_dim=152, out_dim=768):
        super().__init__(resize_input=resize_input, out_dim=out_dim, in_dim=in_dim)
