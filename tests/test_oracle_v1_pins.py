"""Pins for the V1 restatement (oracle/restate_v1.py, oracle/synth_v1.py): the real reference (imported from /root/reference with the
restated timm layers of oracle/stubs/timm) on the same seeded checkpoint -- authoring container only -- and golden vectors
written from that run (tests/golden/v1_*.npz) everywhere else.  What lives in un-vendored dependencies is pinned through statement-by-statement
restatements of the published sources: timm layers (oracle/stubs/timm) and xformers v0.0.26 NystromAttention (oracle/stubs/xformers; the
reference's [b, n, h, d] call takes its plain-softmax branch, tests/test_oracle_nystrom_cpu.py) -- see the header of oracle/restate_v1.py."""
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


def _reference_v1(sd, backbone="cnvnxtl"):
    import contextlib
    import io
    import json
    ref_loader._prepare()
    with warnings.catch_warnings(), contextlib.redirect_stdout(io.StringIO()):
        warnings.simplefilter("ignore")
        from unidepth.models import UniDepthV1  # type: ignore
        cfg = json.load(open(os.path.join(ref_loader.REF_ROOT, "configs", f"config_v1_{backbone}.json")))
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


V1_CASES = {   # name: (B, H, W, seed, intrinsics?, skip_camera)
    "v1_infer_240x320": (1, 240, 320, 5, False, False),
    "v1_infer_200x360_b2_K": (2, 200, 360, 6, True, False),          # aspect padding (pads 15/16 top/bottom) + GT intrinsics
    "v1_infer_200x360_b2_skip": (2, 200, 360, 6, True, True),
}
V1_K = [[250.0, 0.0, 178.0], [0.0, 251.0, 98.0], [0.0, 0.0, 1.0]]


def v1_case_inputs(name):
    B, H, W, seed, withK, skip = V1_CASES[name]
    rgb = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=torch.Generator().manual_seed(seed))
    K = torch.tensor(V1_K).repeat(B, 1, 1) if withK else None
    return rgb, K, skip


def v1_digest(out):
    return {"intrinsics": out["intrinsics"].numpy(), "depth": out["depth"][:, :, 2::5, 3::5].contiguous().numpy(),
            "points": out["points"][:, :, 2::7, 3::7].contiguous().numpy()}


def test_sh_recurrence_is_orthonormal_and_matches_low_orders():
    """real_sh_deg8 (recurrence) -- closed forms of the first bands and Monte-Carlo orthonormality over the sphere."""
    g = torch.Generator().manual_seed(0)
    v = torch.nn.functional.normalize(torch.randn(200000, 3, generator=g, dtype=torch.float64), dim=-1)
    Y = restate_v1.real_sh_deg8(v)
    assert Y.shape == (200000, 81)
    assert torch.allclose(Y[:, 0], torch.full((200000,), 0.282094791773878, dtype=torch.float64))
    assert torch.allclose(Y[:, 1], -0.48860251190292 * v[:, 1]) and torch.allclose(Y[:, 2], 0.48860251190292 * v[:, 2])
    assert torch.allclose(Y[:, 3], -0.48860251190292 * v[:, 0]) and torch.allclose(Y[:, 4], 1.09254843059208 * v[:, 0] * v[:, 1])
    gram = (Y.t() @ Y) * (4 * 3.141592653589793 / 200000)
    assert (gram - torch.eye(81, dtype=torch.float64)).abs().max() < 0.05


@pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not present (GPU box)")
def test_v1_infer_matches_live_reference_with_nystrom_stub():
    """The reference's own UniDepthV1.infer (decoder, pre/post-processing, quirks) with xformers' NystromAttention replaced by the
    oracle's restatement (oracle/stubs/xformers): pins everything of the V1 path except the Nystrom internals."""
    warnings.simplefilter("ignore")
    cfg = synth_v1.load_config_v1()
    sd = synth_v1.make_synthetic_checkpoint_v1(cfg, 211)
    ref = _reference_v1(sd)
    ref.load_state_dict(sd, strict=True)
    from unidepth.utils.sht import rsh_cart_8  # type: ignore
    v = torch.nn.functional.normalize(torch.randn(500, 3, generator=torch.Generator().manual_seed(1)), dim=-1)
    assert (restate_v1.real_sh_deg8(v) - rsh_cart_8(v)).abs().max() < 2e-5
    orc = restate_v1.OracleV1(cfg, sd)
    for name in V1_CASES:
        rgb, K, skip = v1_case_inputs(name)
        with torch.no_grad():
            r = ref.infer(rgb, None if K is None else K.clone(), skip_camera=skip)
        o = orc.infer(rgb, None if K is None else K.clone(), skip_camera=skip)
        for k in r:
            assert (o[k] - r[k]).norm() / r[k].norm() < 2e-5, (name, k)


VITL_CASES = {"v1vitl_infer_200x360_K": (1, 200, 360, 8, True, False), "v1vitl_infer_240x320": (1, 240, 320, 9, False, False)}


def vitl_case_inputs(name):
    B, H, W, seed, withK, skip = VITL_CASES[name]
    rgb = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=torch.Generator().manual_seed(seed))
    return rgb, (torch.tensor(V1_K).repeat(B, 1, 1) if withK else None), skip


@pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not present (GPU box)")
def test_v1_vitl14_matches_live_reference():
    """UniDepthV1 on DINOv2 ViT-L/14 (reference hubconf.py:14-17, configs/config_v1_vitl14.json): strict key set of the synthetic checkpoint,
    the encoder as V1 builds it (interpolate_offset 0.1, no final norm, every block returned, class token added: unidepthv1.py:322-328,
    412-421) and the whole infer() against the real reference (Nystrom stub as for ConvNeXt)."""
    warnings.simplefilter("ignore")
    cfg = synth_v1.load_config_v1("vitl14")
    sd = synth_v1.make_synthetic_checkpoint_v1(cfg, 212)
    ref = _reference_v1(sd, "vitl14")
    missing, unexpected = ref.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    orc = restate_v1.OracleV1(cfg, sd)
    x = _image(1, 126, 168, seed=4)
    with torch.no_grad():
        r_outs, r_cls = ref.pixel_encoder(x)
    outs, cls = orc.encode(x)
    assert len(r_outs) == len(outs) == 24
    for i in (0, 4, 11, 17, 23):
        want = r_outs[i] + r_cls[i].unsqueeze(1)
        assert (outs[i] - want).norm() / want.norm() < TOL, i
        assert (cls[i] - r_cls[i]).norm() / r_cls[i].norm() < TOL, i
    for name in VITL_CASES:
        rgb, K, skip = vitl_case_inputs(name)
        with torch.no_grad():
            r = ref.infer(rgb, None if K is None else K.clone(), skip_camera=skip)
        o = orc.infer(rgb, None if K is None else K.clone(), skip_camera=skip)
        for k in r:
            assert (o[k] - r[k]).norm() / r[k].norm() < 2e-5, (name, k)


@pytest.mark.parametrize("name", list(VITL_CASES))
def test_v1_vitl14_infer_golden(name, golden_dir):
    cfg = synth_v1.load_config_v1("vitl14")
    sd = synth_v1.make_synthetic_checkpoint_v1(cfg, 212)
    rgb, K, skip = vitl_case_inputs(name)
    got = v1_digest(restate_v1.OracleV1(cfg, sd).infer(rgb, K, skip_camera=skip))
    want = np.load(os.path.join(golden_dir, name + ".npz"))
    for k in want.files:
        a, b = got[k].astype(np.float64), want[k].astype(np.float64)
        assert a.shape == b.shape and np.linalg.norm(a - b) / np.linalg.norm(b) < 2e-5, (name, k)


@pytest.mark.parametrize("name", list(V1_CASES))
def test_v1_infer_golden(name, golden_dir):
    cfg = synth_v1.load_config_v1()
    sd = synth_v1.make_synthetic_checkpoint_v1(cfg, 211)
    rgb, K, skip = v1_case_inputs(name)
    got = v1_digest(restate_v1.OracleV1(cfg, sd).infer(rgb, K, skip_camera=skip))
    want = np.load(os.path.join(golden_dir, name + ".npz"))
    for k in want.files:
        a, b = got[k].astype(np.float64), want[k].astype(np.float64)
        assert a.shape == b.shape and np.linalg.norm(a - b) / np.linalg.norm(b) < 2e-5, (name, k)
