"""`UniDepth(version, backbone, pretrained)` entry point with the reference's hubconf signature (hubconf.py:25-41): builds the
engine class for the version from the shipped architecture config and, if `pretrained`, fetches `pytorch_model.bin` from the
`lpiccinelli/unidepth-<version>-<backbone>` hub repository (needs network access or a warm huggingface cache).
The V2 ViT backbones run end to end; `v1` / `cnvnxtl` builds the engine's UniDepthV1 (encoder half implemented, see unidepthv1.py);
`v2old` and the V1 ViT-L variant raise NotImplementedError."""
from __future__ import annotations

import json
import os

import torch

BACKBONES = {"v1": ["vitl14", "cnvnxtl"], "v2": ["vitl14", "vitb14", "vits14"], "v2old": ["vitl14", "vits14"]}
_CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")      # architecture configs of the V2 ViT-S/B/L checkpoints


from .unidepthv1 import UniDepthV1  # noqa: E402,F401  (ConvNeXt-L encoder on the engine; the V1 decoder is still missing and infer() says so)


def UniDepth(version: str = "v2", backbone: str = "vitl14", pretrained: bool = True):
    assert version in BACKBONES, f"version must be one of {list(BACKBONES)}"
    assert backbone in BACKBONES[version], f"backbone for current version ({version}) must be one of {BACKBONES[version]}"
    if version == "v2old" or (version == "v1" and backbone != "cnvnxtl"):
        raise NotImplementedError(f"UniDepth {version}/{backbone} is not implemented on the MI355X engine (SURVEY.md 8f next-1)")
    from .unidepthv2 import UniDepthV2
    with open(os.path.join(_CFG_DIR, f"config_{version}_{backbone}.json")) as f:
        config = json.load(f)
    model = (UniDepthV1 if version == "v1" else UniDepthV2)(config)
    if pretrained:
        import huggingface_hub
        path = huggingface_hub.hf_hub_download(repo_id=f"lpiccinelli/unidepth-{version}-{backbone}", filename="pytorch_model.bin", repo_type="model")
        model.load_state_dict(torch.load(path, map_location="cpu", weights_only=True), strict=False)
    return model
