import torch
import torch.nn as nn
from torch.nn.init import trunc_normal_  # noqa: F401


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        assert not self.training or self.drop_prob == 0.0
        return x


# ---- what backbones/convnext.py needs from timm (ConvNeXt / UniDepthV1 path).  timm is un-vendored and un-pinned in the reference
# (requirements.txt:16), so these are restatements of timm's documented behaviour; ConvNeXt parity is therefore "parity unpinned"
# at this boundary: LayerNorm / LayerNorm2d with eps = 1e-6, Mlp = fc1 -> act -> fc2, create_conv2d with
# symmetric static padding ((stride - 1) + dilation * (k - 1)) // 2 unless a padding is given.
class LayerNorm(nn.LayerNorm):
    def __init__(self, num_channels, eps=1e-6, affine=True):
        super().__init__(num_channels, eps=eps, elementwise_affine=affine)


class LayerNorm2d(nn.LayerNorm):
    """LayerNorm over the channel dimension of an NCHW tensor."""

    def __init__(self, num_channels, eps=1e-6, affine=True):
        super().__init__(num_channels, eps=eps, elementwise_affine=affine)

    def forward(self, x):
        x = x.permute(0, 2, 3, 1)
        x = torch.nn.functional.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        return x.permute(0, 3, 1, 2)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, norm_layer=None, bias=True, drop=0.0,
                 use_conv=False):
        super().__init__()
        assert not use_conv and norm_layer is None and drop == 0.0, "stub: Linear MLP only (what ConvNeXt(conv_mlp=False) builds)"
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


def _unavailable(name):
    class _U(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"timm.{name} is not available in the oracle stub")
    _U.__name__ = name
    return _U


AvgPool2dSame = _unavailable("AvgPool2dSame")
GlobalResponseNormMlp = _unavailable("GlobalResponseNormMlp")


def create_conv2d(in_channels, out_channels, kernel_size, stride=1, dilation=1, depthwise=False, bias=True, padding="", **kwargs):
    assert not kwargs, kwargs
    if padding == "":
        padding = ((stride - 1) + dilation * (kernel_size - 1)) // 2
    assert isinstance(padding, int), "stub: static symmetric padding only"
    groups = in_channels if depthwise else 1
    return nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation, groups=groups, bias=bias)


def get_act_layer(name):
    return {"gelu": nn.GELU, "relu": nn.ReLU}[name]


def make_divisible(v, divisor=8, min_value=None, round_limit=0.9):
    min_value = min_value or divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


def to_ntuple(n):
    def parse(x):
        if isinstance(x, (tuple, list)):
            return tuple(x)
        return tuple([x] * n)
    return parse
