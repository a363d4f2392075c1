import torch as th


class _GroundingNetInputBase:
    """prepare() forwards the listed batch entries (optionally renamed); get_null_input() returns
    all-zero tensors of the shapes seen by the last prepare() — the reference's null-guidance input."""

    fields = ()       # ((batch key, PositionNet kwarg), ...)
    shape_key = None  # batch entry that defines (batch, n, feature dim)

    def __init__(self):
        self.set = False
        self._shapes = {}

    def prepare(self, batch):
        self.set = True
        ref = batch[self.shape_key]
        self.batch = ref.shape[0]
        self.device, self.dtype = ref.device, ref.dtype
        out = {}
        for src, dst in self.fields:
            out[dst] = batch[src]
            self._shapes[dst] = tuple(batch[src].shape[1:])
        self._remember(ref)
        return out

    def _remember(self, ref):
        pass

    def get_null_input(self, batch=None, device=None, dtype=None):
        assert self.set, "not set yet, cannot call this funcion"
        batch = self.batch if batch is None else batch
        device = self.device if device is None else device
        dtype = self.dtype if dtype is None else dtype
        return {k: th.zeros((batch, *s), dtype=dtype, device=device) for k, s in self._shapes.items()}


class _SpatialNetInputBase:
    """Spatial-map tokenizers (reference grounding_input/{canny,hed,depth,normal,sem}_grounding_tokinzer_input.py): prepare()
    forwards (image, mask); the null input is an all-zero image of the last seen batch and a zero mask."""

    image_key = None

    def __init__(self):
        self.set = False

    def prepare(self, batch):
        self.set = True
        img, mask = batch[self.image_key], batch["mask"]
        self.batch, self.C, self.H, self.W = img.shape
        self.device, self.dtype = img.device, img.dtype
        return {self.image_key: img, "mask": mask}

    def get_null_input(self, batch=None, device=None, dtype=None):
        assert self.set, "not set yet, cannot call this funcion"
        batch = self.batch if batch is None else batch
        device = self.device if device is None else device
        dtype = self.dtype if dtype is None else dtype
        # as in the reference: the zero image keeps the batch seen by prepare(), only the mask follows `batch`
        return {self.image_key: th.zeros(self.batch, self.C, self.H, self.W).type(dtype).to(device),
                "mask": th.zeros(batch).type(dtype).to(device)}
