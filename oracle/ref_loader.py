"""TEST INFRASTRUCTURE ONLY -- import the *real* reference (read-only tree at /root/reference) on CPU.

The reference imports a few packages that are not installed here (torchvision, timm, cv2, wandb); the
tiny stand-ins under oracle/stubs/ satisfy the imports (SURVEY.md section 8c lists what each one must
provide).  /root/reference does not exist on the GPU box: callers must check `available()` first and
nothing in `-m gpu` tests, smoke() or bench.py may depend on this module.
"""
from __future__ import annotations

import json
import os
import sys
import warnings

REF_ROOT = "/root/reference"
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "unidepth"))


def _prepare():
    sys.dont_write_bytecode = True            # reference tree is read-only
    for p in (REF_ROOT, _STUBS):
        if p not in sys.path:
            sys.path.insert(0, p)


def reference_config(name: str) -> dict:
    with open(os.path.join(REF_ROOT, "configs", f"config_v2_{name}.json")) as f:
        return json.load(f)


def build_reference(name: str, state_dict=None):
    """Instantiate the reference UniDepthV2 (fp32, CPU, eval) and strictly load `state_dict`."""
    assert available(), "reference tree not present"
    _prepare()
    import contextlib
    import io

    with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
        warnings.simplefilter("ignore")
        from unidepth.models import UniDepthV2  # type: ignore

        model = UniDepthV2(reference_config(name)).eval()
    if state_dict is not None:
        missing, unexpected = model.load_state_dict(state_dict, strict=True)
        assert not missing and not unexpected
    return model
