"""Pure-tensor entry points with the signatures of the reference's ONNX wrappers (unidepth/models/unidepthv2/export.py:18-76):
`UniDepthV2ONNX.forward(rgbs)` and `UniDepthV2ONNXcam.forward(rgbs, rays)` take the NETWORK image (already normalised, H and W
multiples of 14; no aspect padding / resize / post-processing) and return (pts_3d [B,3,H,W], confidence [B,1,H,W],
intrinsics [B,3,3]).  Here they run the engine's launch program through the pixel_encoder / pixel_decoder seams; there is no
ONNX graph to export -- the engine IS the deployment artefact on MI355X (the reference exports to ONNX to leave PyTorch)."""
from __future__ import annotations

from .unidepthv2 import UniDepthV2


class UniDepthV2ONNX(UniDepthV2):
    def forward(self, rgbs):
        return self.forward_export(rgbs)

    __call__ = forward


class UniDepthV2ONNXcam(UniDepthV2):
    def forward(self, rgbs, rays):
        return self.forward_export(rgbs, rays)

    __call__ = forward
