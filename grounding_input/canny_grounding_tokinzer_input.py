"""canny map grounding input (reference grounding_input/canny_grounding_tokinzer_input.py)."""
from grounding_input._base import _SpatialNetInputBase


class GroundingNetInput(_SpatialNetInputBase):
    image_key = "canny_edge"
