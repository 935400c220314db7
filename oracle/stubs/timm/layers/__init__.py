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


def _unavailable(name):
    class _U(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"timm.{name} is not available in the oracle stub (ConvNeXt/V1 path)")
    _U.__name__ = name
    return _U


AvgPool2dSame = _unavailable("AvgPool2dSame")
GlobalResponseNormMlp = _unavailable("GlobalResponseNormMlp")
LayerNorm = _unavailable("LayerNorm")
LayerNorm2d = _unavailable("LayerNorm2d")
Mlp = _unavailable("Mlp")


def create_conv2d(*a, **k):
    raise NotImplementedError("timm.create_conv2d not available in the oracle stub")


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
