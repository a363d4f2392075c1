"""Box + text grounding input (reference grounding_input/text_grounding_tokinzer_input.py; the
misspelt module name is part of the checkpoint/config contract)."""
from grounding_input._base import _GroundingNetInputBase


class GroundingNetInput(_GroundingNetInputBase):
    fields = (("boxes", "boxes"), ("masks", "masks"), ("text_embeddings", "positive_embeddings"))
    shape_key = "text_embeddings"

    def _remember(self, ref):
        _, self.max_box, self.in_dim = ref.shape
