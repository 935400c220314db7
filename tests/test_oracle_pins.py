"""Pins for the CPU oracle (oracle/restate.py):
  (1) against the golden vectors produced by the REAL reference (tests/golden/*.npz, made by
      oracle/make_golden.py) -- runs everywhere, incl. the GPU box;
  (2) against the real reference imported live from /root/reference -- authoring container only.
Tolerance: fp32 round-off (the restatement executes the same ATen CPU kernels; observed 0 .. 4e-7)."""
import os
import warnings

import numpy as np
import pytest
import torch

from oracle import cases, ref_loader, restate, synth

TOL = 5e-6   # rel-L2 on each digest entry


def _run_oracle(case):
    cfg = synth.load_config(case["arch"])
    sd = synth.make_synthetic_checkpoint(cfg, case["ckpt_seed"])
    orc = restate.OracleV2(cfg, sd)
    if case.get("resolution_level") is not None:
        orc.resolution_level = case["resolution_level"]
    rgb, cam = cases.case_inputs(case)
    return orc.infer(rgb, cam), sd


@pytest.mark.parametrize("name", list(cases.CASES))
def test_oracle_matches_reference_golden(name, golden_dir):
    case = cases.CASES[name]
    out, _ = _run_oracle(case)
    got = cases.digest(out)
    want = np.load(os.path.join(golden_dir, name + ".npz"))
    assert set(got) == set(want.files)
    for k in want.files:
        a, b = got[k].astype(np.float64), want[k].astype(np.float64)
        assert a.shape == b.shape, (k, a.shape, b.shape)
        rel = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
        assert rel < TOL, f"{name}:{k} rel-l2 {rel:.3e}"
    # the sensitised checkpoint must make depth input dependent: sanity on spread
    assert out["depth"].std() / out["depth"].mean() > 0.05


@pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not present (GPU box)")
def test_oracle_matches_live_reference_and_keys():
    warnings.simplefilter("ignore")
    case = dict(arch="vits14", H=322, W=490, B=2, camera=False, ckpt_seed=77, img_seed=9)
    out, sd = _run_oracle(case)
    ref = ref_loader.build_reference(case["arch"], sd)      # strict=True: key set/shapes of synth == reference
    rgb, cam = cases.case_inputs(case)
    with torch.no_grad():
        rout = ref.infer(rgb, cam)
    assert set(rout) == set(out)
    for k in rout:
        a, b = out[k].double(), rout[k].double()
        assert a.shape == b.shape
        assert (a - b).norm() / b.norm() < TOL, k


def test_shape_policy_table():
    """SURVEY.md 8d shape-policy table [probe] (reference unidepthv2.py:36-77)."""
    sc = synth.load_config("vitl14")["data"]["augmentations"]["shape_constraints"]
    table = {(518, 518): (518, 518), (644, 966): (644, 952), (462, 616): (462, 616), (480, 640): (490, 644),
             (900, 1600): (588, 1036), (375, 1242): (490, 1232)}
    for (H, W), want in table.items():
        pads, padded = restate.get_paddings((H, W), sc["ratio_bounds"])
        _, got = restate.get_resize_factor(padded, (sc["pixels_min"], sc["pixels_max"]))
        assert got == want, ((H, W), got, want)
    assert restate.get_paddings((375, 1242), sc["ratio_bounds"])[0] == (0, 0, 60, 61)
