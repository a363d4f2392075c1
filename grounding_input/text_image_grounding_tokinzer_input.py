"""Box + text + image grounding input (reference grounding_input/text_image_grounding_tokinzer_input.py)."""
from grounding_input._base import _GroundingNetInputBase


class GroundingNetInput(_GroundingNetInputBase):
    fields = tuple((k, k) for k in ("boxes", "masks", "text_masks", "image_masks", "text_embeddings", "image_embeddings"))
    shape_key = "text_embeddings"

    def _remember(self, ref):
        _, self.max_box, self.in_dim = ref.shape
