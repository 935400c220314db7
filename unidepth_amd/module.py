"""`torch.nn.Module` surface of the engine classes.

The reference declares `class UniDepthV2(nn.Module, PyTorchModelHubMixin, library_name="UniDepth", ...)`
(unidepth/models/unidepthv2/unidepthv2.py:111-117; V1: unidepthv1/unidepthv1.py:97-103).  Code written against it may type-check the model
(`isinstance(model, nn.Module)`), call it (`model(rgb)`), walk `.parameters()` / `.modules()`, cast it (`.half()`), or `push_to_hub`.
The engine holds its weights as repacked device buffers behind the C-ABI, not as `nn.Parameter`s, so this base gives those calls a defined
meaning instead of an AttributeError:

* `isinstance(model, torch.nn.Module)` holds; `parameters()` / `buffers()` are empty (nothing is trainable here: inference engine);
* `model(...)` = `forward(...)` = `infer(...)`;
* `.half() / .float() / .double() / .bfloat16() / .to(dtype)` are no-ops with a warning: operand precision is part of the kernels
  (fp16 MFMA operands, fp32 statistics and residual streams), not a property of stored tensors;
* `.train()` raises for `mode=True` (there is no backward), `.eval()` / `.train(False)` return self, `requires_grad_()` is a no-op;
* `state_dict()` returns the fp32 dict with the REFERENCE's key names (what was loaded), `load_state_dict` keeps the engine's chaining
  return value (`model.load_state_dict(sd).to("cuda")`);
* with huggingface_hub importable the classes also derive from `PyTorchModelHubMixin`, so `push_to_hub` exists and goes through the
  engine's own `save_pretrained` (config.json + model.safetensors, the layout `from_pretrained` of both implementations reads).
"""
from __future__ import annotations

import warnings

import torch

try:                                                    # the reference's second base class; optional here
    from huggingface_hub import PyTorchModelHubMixin as _HubMixin
except Exception:                                       # pragma: no cover - huggingface_hub missing
    class _HubMixin:                                    # type: ignore[no-redef]
        def __init_subclass__(cls, **kwargs):
            super().__init_subclass__()


class EngineModule(torch.nn.Module, _HubMixin):
    """Common nn.Module plumbing of UniDepthV1 / UniDepthV2 (see the module docstring)."""

    def __init__(self):
        super().__init__()
        self.training = False
        self._device = torch.device("cpu")

    # ---- device / dtype ------------------------------------------------------------------------------------------------
    @property
    def device(self):
        return self._device

    def _move(self, device: torch.device) -> None:       # subclasses drop their packed weights / plans here
        raise NotImplementedError

    def to(self, *args, **kwargs):
        device = kwargs.get("device")
        dtype = kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            elif isinstance(a, bool):
                continue                                    # positional non_blocking (nn.Module.to(device, dtype, non_blocking)): nothing to do
            elif isinstance(a, (str, torch.device, int)):
                device = a
            elif isinstance(a, torch.Tensor):
                device, dtype = a.device, a.dtype
        if dtype is not None:
            self._dtype_noop(f"to({dtype})")
        if device is not None:
            device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
            if device.type == "cuda" and device.index is None:
                device = torch.device("cuda", torch.cuda.current_device())
            if device != self._device:
                self._device = device
                self._move(device)
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def cpu(self):
        return self.to("cpu")                             # allowed (weights stay on the host anyway); infer() then raises: no CPU path

    def _dtype_noop(self, what: str):
        warnings.warn(f"{type(self).__name__}.{what}: no-op -- the MI355X engine fixes operand precision inside its kernels "
                      "(fp16 MFMA operands, fp32 accumulation / statistics); stored weights stay fp32 and are repacked on .to('cuda')")
        return self

    def half(self):
        return self._dtype_noop("half()")

    def float(self):
        return self._dtype_noop("float()")

    def double(self):
        return self._dtype_noop("double()")

    def bfloat16(self):
        return self._dtype_noop("bfloat16()")

    # ---- mode ------------------------------------------------------------------------------------------------------------
    def train(self, mode: bool = True):
        if mode:
            raise RuntimeError(f"{type(self).__name__} is an inference engine (forward kernels only): train(True) is not available")
        self.training = False
        return self

    def eval(self):
        return self.train(False)

    def requires_grad_(self, requires_grad: bool = True):
        return self

    # ---- call path -------------------------------------------------------------------------------------------------------
    def forward(self, *args, **kwargs):
        """`model(rgb, camera)` runs infer(): the reference's training-time forward(inputs, image_metas) has no counterpart here."""
        return self.infer(*args, **kwargs)

    # ---- weights ---------------------------------------------------------------------------------------------------------
    def state_dict(self, *args, **kwargs):
        """fp32 tensors under the reference's key names (exactly what load_state_dict received, `module.` prefixes stripped)."""
        sd = getattr(self, "_sd", None)
        return dict(sd) if sd is not None else {}

    def save_pretrained(self, save_directory, *, config=None, repo_id=None, push_to_hub: bool = False, **push_to_hub_kwargs):
        """HF layout (config.json + model.safetensors); the keyword arguments are the ones PyTorchModelHubMixin.push_to_hub passes."""
        import json
        import os
        from safetensors.torch import save_file
        save_directory = str(save_directory)
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(self.config, f)
        save_file({k: v.contiguous() for k, v in self._sd.items()}, os.path.join(save_directory, "model.safetensors"))
        if push_to_hub and hasattr(super(), "push_to_hub"):
            kw = dict(push_to_hub_kwargs)
            return super().push_to_hub(repo_id=repo_id or os.path.basename(save_directory.rstrip("/")), **kw)
        return None
