"""canny map as the UNet's grounding_extra_input (reference grounding_input/canny_grounding_downsampler_input.py)."""


class GroundingDSInput:
    def prepare(self, batch):
        return batch["canny_edge"]
