"""unidepth_amd -- MI355X (gfx950) native engine for the UniDepth `infer()` paths (UniDepthV2 on DINOv2 ViT-S/B/L; UniDepthV1 on ConvNeXt-L).

Public surface mirrors the reference (lpiccinelli-eth/UniDepth, unidepth/models/__init__.py):
    from unidepth_amd import UniDepthV2
    model = UniDepthV2.from_pretrained(dir_or_repo).to("cuda").eval(); out = model.infer(rgb, camera)
All device arithmetic runs in libunidepth_hip.so (hand-written HIP); importing this package without the
built library raises ImportError -- there is no CPU / eager-PyTorch fallback."""
from . import _lib  # noqa: F401  (fails loudly when the HIP library is missing)

__all__ = ["UniDepthV2", "UniDepthV1", "UniDepth"]


def __getattr__(name):
    if name == "UniDepthV2":
        from .unidepthv2 import UniDepthV2
        return UniDepthV2
    if name in ("UniDepthV1", "UniDepth"):            # hubconf-style entry point / the V1 family (ConvNeXt-L / ViT-L)
        from . import hub
        return getattr(hub, name)
    raise AttributeError(name)
