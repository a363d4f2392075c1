"""ConvNeXt backbone of the spatial-map grounding tokenizers — parameter container with the state_dict keys of the
reference's ldm/modules/diffusionmodules/convnext.py (downsample_layers.{0..3}.{0,1}.*, stages.{i}.{j}.{dwconv,norm,
pwconv1,pwconv2}.*, stages.{i}.{j}.gamma; the classifier head is removed there too, convnext.py:99-104). It runs once
per prompt, in front of the denoising loop."""
import torch
import torch.nn as nn

from ldm.modules.attention import _EngineOnly


class LayerNorm(_EngineOnly):
    """channels_last / channels_first LayerNorm (convnext.py:123-147): weight, bias, eps."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        super().__init__()
        if data_format not in ("channels_last", "channels_first"):
            raise NotImplementedError
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps, self.data_format, self.normalized_shape = eps, data_format, (normalized_shape,)


class Block(_EngineOnly):
    """dwconv 7x7 -> LayerNorm -> Linear(4x) -> GELU -> Linear -> gamma * x, + input (convnext.py:15-50)."""

    def __init__(self, dim, drop_path=0.0, layer_scale_init_value=1e-6):
        super().__init__()
        if drop_path:
            raise NotImplementedError("stochastic depth is a training feature")
        self.dwconv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones(dim)) if layer_scale_init_value > 0 else None


class ConvNeXt(_EngineOnly):
    def __init__(self, in_chans=3, num_classes=1000, depths=(3, 3, 9, 3), dims=(96, 192, 384, 768), drop_path_rate=0.0,
                 layer_scale_init_value=1e-6, head_init_scale=1.0):
        super().__init__()
        self.depths, self.dims = tuple(depths), tuple(dims)
        self.downsample_layers = nn.ModuleList([nn.Sequential(nn.Conv2d(in_chans, dims[0], kernel_size=4, stride=4),
                                                              LayerNorm(dims[0], eps=1e-6, data_format="channels_first"))])
        for i in range(3):
            self.downsample_layers.append(nn.Sequential(LayerNorm(dims[i], eps=1e-6, data_format="channels_first"),
                                                        nn.Conv2d(dims[i], dims[i + 1], kernel_size=2, stride=2)))
        self.stages = nn.ModuleList([nn.Sequential(*[Block(dims[i], layer_scale_init_value=layer_scale_init_value) for _ in range(depths[i])])
                                     for i in range(4)])


def convnext_tiny(pretrained=False, in_22k=False, **kwargs):
    """The reference downloads ImageNet weights here when pretrained=True (convnext.py:160-166); GLIGEN checkpoints
    carry the fine-tuned backbone in their state_dict, so nothing is fetched: load_state_dict fills it."""
    return ConvNeXt(depths=[3, 3, 9, 3], dims=[96, 192, 384, 768], **kwargs)
