"""`UniDepth(version, backbone, pretrained)` entry point with the reference's hubconf signature (hubconf.py:25-41): builds the
engine class for the version from the shipped architecture config and, if `pretrained`, fetches `pytorch_model.bin` from the
`lpiccinelli/unidepth-<version>-<backbone>` hub repository (needs network access or a warm huggingface cache).
The V2 ViT backbones and both V1 variants (`v1` / `cnvnxtl`: ConvNeXt-L; `v1` / `vitl14`: DINOv2 ViT-L/14; unidepthv1.py) run end to end on the
engine; `v2old` raises NotImplementedError.

UniDepthV1's `layers_8` / `layers_4` (NystromBlock, layers/nystrom_attention.py) are computed as the reference's call into xformers'
NystromAttention computes them for its 4-D [b, n, h, d] tensors -- the module's small-sequence branch, a per-token softmax attention among the h
head-vectors (oracle/stubs/xformers restates the module; unidepth_amd/unidepthv1.py nystrom_block) -- not as the paper's landmark algorithm."""
from __future__ import annotations

import json
import os

import torch

BACKBONES = {"v1": ["vitl14", "cnvnxtl"], "v2": ["vitl14", "vitb14", "vits14"], "v2old": ["vitl14", "vits14"]}
_CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")      # architecture configs of the V2 ViT-S/B/L checkpoints


from .unidepthv1 import UniDepthV1  # noqa: E402,F401  (ConvNeXt-L encoder + V1 decoder on the engine)


def UniDepth(version: str = "v2", backbone: str = "vitl14", pretrained: bool = True):
    assert version in BACKBONES, f"version must be one of {list(BACKBONES)}"
    assert backbone in BACKBONES[version], f"backbone for current version ({version}) must be one of {BACKBONES[version]}"
    if version == "v2old":
        raise NotImplementedError(f"UniDepth {version}/{backbone} is not implemented on the MI355X engine (SURVEY.md 2: out of scope)")
    from .unidepthv2 import UniDepthV2
    with open(os.path.join(_CFG_DIR, f"config_{version}_{backbone}.json")) as f:
        config = json.load(f)
    model = (UniDepthV1 if version == "v1" else UniDepthV2)(config)
    if pretrained:
        import huggingface_hub
        path = huggingface_hub.hf_hub_download(repo_id=f"lpiccinelli/unidepth-{version}-{backbone}", filename="pytorch_model.bin", repo_type="model")
        model.load_state_dict(torch.load(path, map_location="cpu", weights_only=True), strict=False)
    return model
