"""`UniDepth(version, backbone, pretrained)` entry point with the reference's hubconf signature (hubconf.py:25-41): builds the
engine class for the version from the shipped architecture config and, if `pretrained`, fetches `pytorch_model.bin` from the
`lpiccinelli/unidepth-<version>-<backbone>` hub repository (needs network access or a warm huggingface cache).
Only the V2 ViT backbones run on this engine; `v1` / `v2old` raise NotImplementedError (see UniDepthV1 below)."""
from __future__ import annotations

import json
import os

import torch

BACKBONES = {"v1": ["vitl14", "cnvnxtl"], "v2": ["vitl14", "vitb14", "vits14"], "v2old": ["vitl14", "vits14"]}
_CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")      # architecture configs of the V2 ViT-S/B/L checkpoints


class UniDepthV1:
    """Placeholder for the reference's V1 model family (unidepth/models/unidepthv1/unidepthv1.py:288-373; ConvNeXt-L / ViT-L
    encoders, spherical-harmonics camera embedding, Nystrom attention decoder).  Not built on this engine yet: its decoder's
    arithmetic lives in an un-vendored, un-pinned third-party module (xformers NystromAttention), so no oracle can be pinned
    (SURVEY.md 8c / 8f next-1).  Constructing it fails loudly instead of silently running something else."""

    def __init__(self, *a, **kw):
        raise NotImplementedError("UniDepthV1 is not implemented on the MI355X engine (SURVEY.md 8f next-1); use UniDepthV2")

    @classmethod
    def from_pretrained(cls, *a, **kw):
        return cls()


def UniDepth(version: str = "v2", backbone: str = "vitl14", pretrained: bool = True):
    assert version in BACKBONES, f"version must be one of {list(BACKBONES)}"
    assert backbone in BACKBONES[version], f"backbone for current version ({version}) must be one of {BACKBONES[version]}"
    if version != "v2":
        raise NotImplementedError(f"UniDepth {version} is not implemented on the MI355X engine (SURVEY.md 8f next-1); use version='v2'")
    from .unidepthv2 import UniDepthV2
    with open(os.path.join(_CFG_DIR, f"config_v2_{backbone}.json")) as f:
        config = json.load(f)
    model = UniDepthV2(config)
    if pretrained:
        import huggingface_hub
        path = huggingface_hub.hf_hub_download(repo_id=f"lpiccinelli/unidepth-{version}-{backbone}", filename="pytorch_model.bin", repo_type="model")
        model.load_state_dict(torch.load(path, map_location="cpu", weights_only=True), strict=False)
    return model
