"""Keypoint grounding input (reference grounding_input/keypoint_grounding_tokinzer_input.py)."""
from grounding_input._base import _GroundingNetInputBase


class GroundingNetInput(_GroundingNetInputBase):
    fields = (("points", "points"), ("masks", "masks"))
    shape_key = "points"

    def _remember(self, ref):
        self.max_persons_per_image = int(ref.shape[1] / 17)
