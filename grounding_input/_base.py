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
