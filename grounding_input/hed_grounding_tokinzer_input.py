"""hed map grounding input (reference grounding_input/hed_grounding_tokinzer_input.py)."""
from grounding_input._base import _SpatialNetInputBase


class GroundingNetInput(_SpatialNetInputBase):
    image_key = "hed_edge"
