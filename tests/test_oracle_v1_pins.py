"""Pins for the V1 restatement (oracle/restate_v1.py, oracle/synth_v1.py): the real reference (imported from /root/reference with the
restated timm layers of oracle/stubs/timm) on the same seeded checkpoint -- authoring container only -- and golden vectors
written from that run (tests/golden/v1_*.npz) everywhere else.  PARITY UNPINNED for what lives in un-vendored dependencies
(timm layer semantics, xformers NystromAttention): see the header of oracle/restate_v1.py."""
import os
import sys
import warnings

import numpy as np
import pytest
import torch

from oracle import ref_loader, restate_v1, synth_v1

TOL = 5e-6


def _image(B=1, H=128, W=160, seed=3):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 3, H, W, generator=g)


def _reference_v1(sd):
    import contextlib
    import io
    import json
    ref_loader._prepare()
    with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
        warnings.simplefilter("ignore")
        from unidepth.models import UniDepthV1  # type: ignore
        cfg = json.load(open(os.path.join(ref_loader.REF_ROOT, "configs", "config_v1_cnvnxtl.json")))
        model = UniDepthV1(cfg).eval()
    return model


@pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not present (GPU box)")
def test_v1_key_set_and_convnext_encoder_match_live_reference():
    warnings.simplefilter("ignore")
    cfg = synth_v1.load_config_v1()
    sd = synth_v1.make_synthetic_checkpoint_v1(cfg, 211)
    ref = _reference_v1(sd)
    missing, unexpected = ref.load_state_dict(sd, strict=True)          # key set / shapes of synth_v1 == reference
    assert not missing and not unexpected
    x = _image()
    with torch.no_grad():
        r_outs, r_cls = ref.pixel_encoder(x)
    orc = restate_v1.OracleConvNeXt(cfg, sd)
    outs, cls = orc.encode(x)
    assert len(outs) == len(r_outs) == 36
    for i in (0, 2, 3, 5, 6, 20, 32, 33, 35):
        assert outs[i].shape == r_outs[i].shape
        assert (outs[i] - r_outs[i]).norm() / r_outs[i].norm() < TOL, i
        assert (cls[i] - r_cls[i]).norm() / r_cls[i].norm() < TOL, i
    # sensitised: the deepest features must depend on the input
    outs2, _ = orc.encode(_image(seed=4))
    assert (outs2[35] - outs[35]).norm() / outs[35].norm() > 0.05


def test_v1_encoder_golden(golden_dir):
    """Golden digest written by oracle/make_golden_v1.py from the REAL reference encoder (restated timm layers)."""
    cfg = synth_v1.load_config_v1()
    sd = synth_v1.make_synthetic_checkpoint_v1(cfg, 211, encoder_only=True)
    orc = restate_v1.OracleConvNeXt(cfg, sd)
    outs, cls = orc.encode(_image())
    feats = orc.stage_features(outs)
    want = np.load(os.path.join(golden_dir, "v1_convnext_128x160.npz"))
    for j in range(4):
        a, b = feats[j][:, ::3, ::3, ::7].numpy().astype(np.float64), want[f"stage{j}"].astype(np.float64)
        assert a.shape == b.shape
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < TOL, j
    a, b = torch.cat([cls[-i - 1] for i in range(4)], dim=-1).numpy(), want["cls_last4"]
    assert np.linalg.norm(a - b) / np.linalg.norm(b) < TOL
