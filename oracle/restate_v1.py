"""TEST INFRASTRUCTURE ONLY -- CPU fp32 restatement of the UniDepthV1 inference path (BASELINE.json configs[3]).
Nothing under unidepth_amd/ imports this file; it is the checker for the HIP engine's V1 components.

Encoder = ConvNeXt-L as the reference builds it (unidepth/models/encoder.py:127-136 -> backbones/convnext.py:301-471):
  stem Conv2d(3, 192, k=4, s=4) + LayerNorm2d                                 convnext.py:370-383
  4 stages, depths (3, 3, 27, 3), dims (192, 384, 768, 1536); stages 1..3 start with LayerNorm2d + Conv2d(k=2, s=2)   :245-266
  block: depth-wise 7x7 conv (pad 3) -> LayerNorm (channels last) -> Linear(C, 4C) -> GELU(erf) -> Linear(4C, C) -> * gamma -> + input   :130-223
  forward returns EVERY block's output in NHWC plus its spatial mean as a "class token"                           :447-458
LayerNorm / LayerNorm2d / Mlp / create_conv2d come from timm (un-vendored, un-pinned: requirements.txt:16): eps = 1e-6, symmetric
static padding.  PARITY UNPINNED at that boundary: no reference test or golden vector covers the ConvNeXt path; the pins of this
file are the reference's own code executed with the restated timm layers of oracle/stubs/timm (tests/test_oracle_v1_pins.py).
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.nn.functional as F

CONVNEXT = {"convnext_large": dict(depths=(3, 3, 27, 3), dims=(192, 384, 768, 1536))}   # encoder.py:127-136


def convnext_arch(config: dict) -> dict:
    a = dict(CONVNEXT[config["model"]["pixel_encoder"]["name"]])
    a["output_idx"] = list(config["model"]["pixel_encoder"].get("output_idx", [3, 6, 33, 36]))   # cumulative stage ends
    return a


class OracleConvNeXt:
    """Functional ConvNeXt over a dict of fp32 tensors keyed like the reference state_dict (prefix `pixel_encoder.`)."""

    def __init__(self, config: dict, state_dict: dict, prefix: str = "pixel_encoder."):
        self.a = convnext_arch(config)
        self.p = prefix
        self.w = {k: v.detach().to(torch.float32).cpu() for k, v in state_dict.items() if k.startswith(prefix)}
        self.taps: dict = {}
        self.keep_taps = False

    def _ln2d(self, x, name):                      # LayerNorm2d: over C of NCHW, eps 1e-6 (timm)
        y = F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), self.w[name + ".weight"], self.w[name + ".bias"], 1e-6)
        return y.permute(0, 3, 1, 2)

    def block(self, x, name):
        """ConvNeXtBlock.forward, conv_mlp=False (convnext.py:206-223)."""
        w = self.w
        C = x.shape[1]
        y = F.conv2d(x, w[name + ".conv_dw.weight"], w[name + ".conv_dw.bias"], padding=3, groups=C)
        y = y.permute(0, 2, 3, 1)
        y = F.layer_norm(y, (C,), w[name + ".norm.weight"], w[name + ".norm.bias"], 1e-6)
        y = F.linear(F.gelu(F.linear(y, w[name + ".mlp.fc1.weight"], w[name + ".mlp.fc1.bias"])), w[name + ".mlp.fc2.weight"], w[name + ".mlp.fc2.bias"])
        y = y.permute(0, 3, 1, 2) * w[name + ".gamma"].reshape(1, -1, 1, 1)
        return y + x

    def encode(self, image: torch.Tensor) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
        """ConvNeXt.forward (convnext.py:447-458): image [B,3,H,W] (normalised) -> (36 block outputs [B,h,w,C], 36 means [B,1,C])."""
        p, w = self.p, self.w
        x = F.conv2d(image, w[p + "stem.0.weight"], w[p + "stem.0.bias"], stride=4)
        x = self._ln2d(x, p + "stem.1")
        if self.keep_taps:
            self.taps["stem"] = x.clone()
        outs = []
        for s, depth in enumerate(self.a["depths"]):
            if s > 0:
                x = self._ln2d(x, f"{p}stages.{s}.downsample.0")
                x = F.conv2d(x, w[f"{p}stages.{s}.downsample.1.weight"], w[f"{p}stages.{s}.downsample.1.bias"], stride=2)
            for b in range(depth):
                x = self.block(x, f"{p}stages.{s}.blocks.{b}")
                outs.append(x.permute(0, 2, 3, 1).contiguous())
        return outs, [o.mean(dim=(1, 2)).unsqueeze(1).contiguous() for o in outs]

    def stage_features(self, outs: List[torch.Tensor]) -> List[torch.Tensor]:
        """What the V1 decoder consumes (unidepthv1/decoder.py:366-373, 468-480): per stage the element-wise MAX over the stage's
        block outputs (utils/misc.py:18-21 max_stack over slices_encoder_range)."""
        ends = self.a["output_idx"]
        starts = [0] + ends[:-1]
        feats = []
        for i, j in zip(starts, ends):
            feats.append(outs[i] if j - i == 1 else torch.stack(outs[i:j], dim=-1).max(dim=-1).values)
        return feats
