"""TEST INFRASTRUCTURE ONLY -- writes tests/golden/eval_*.npz from the REFERENCE run in the authoring container:
  * knn: the reference's Python wrapper (unidepth/ops/knn/functions/knn.py knn_points) over its own CPU extension compiled from
    its sources (oracle/build_ref_knn.py), and the reference ChamferDistance (utils/chamfer_distance.py) on top of it;
  * patches: the reference's pure-torch twin of its CUDA kernel (unidepth/ops/losses/local_ssi.py:44-77 extract_patches).
Inputs are regenerated from seeds by eval_cases() on any box; only the reference OUTPUTS are stored.

    python -m oracle.make_golden_eval"""
from __future__ import annotations

import importlib.util
import os
import sys
import warnings

import numpy as np
import torch

from . import build_ref_knn, ref_loader

_HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(_HERE), "tests", "golden")


def knn_case_inputs(name: str):
    """Seeded inputs of the K-NN golden cases: (p1, p2, lengths1, lengths2, norm, K)."""
    spec = {
        "d3_k1": (5, 2, 700, 900, 3, 1, 2, False),
        "d3_k5_ragged": (6, 3, 300, 257, 3, 5, 2, True),
        "d2_k3_l1": (7, 2, 130, 64, 2, 3, 1, True),
        "d8_k8": (8, 1, 200, 333, 8, 8, 2, False),
        "d16_k4": (9, 2, 100, 150, 16, 4, 2, True),
        "d3_k4_ties": (10, 2, 256, 300, 3, 4, 2, False),        # integer lattice: many exactly tied distances
        "d3_k6_short": (11, 3, 40, 9, 3, 6, 2, True),           # clouds with fewer than K points
    }[name]
    seed, N, P1, P2, D, K, norm, ragged = spec
    g = torch.Generator().manual_seed(seed)
    if name.endswith("ties"):
        p1 = torch.randint(0, 4, (N, P1, D), generator=g).float()
        p2 = torch.randint(0, 4, (N, P2, D), generator=g).float()
    else:
        p1 = torch.randn(N, P1, D, generator=g)
        p2 = torch.randn(N, P2, D, generator=g)
    l1 = l2 = None
    if ragged:
        l1 = torch.randint(P1 // 2, P1 + 1, (N,), generator=g)
        l2 = torch.randint(max(1, P2 // 3), P2 + 1, (N,), generator=g)
        if name.endswith("short"):
            l2 = torch.tensor([2, 9, 0][:N])
            l1[0] = 0
    return p1, p2, l1, l2, norm, K


KNN_CASES = ["d3_k1", "d3_k5_ragged", "d2_k3_l1", "d8_k8", "d16_k4", "d3_k4_ties", "d3_k6_short"]


def patch_case_inputs(name: str):
    """(tensor [B,1,H,W], centers [B,N,2] (y, x), patch_size (w, h))."""
    spec = {"p7x5": (21, 2, 37, 53, 11, (7, 5)), "p9": (22, 1, 96, 128, 6, (9, 9)), "p3_border": (23, 2, 16, 20, 9, (3, 3))}[name]
    seed, B, H, W, N, ps = spec
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(B, 1, H, W, generator=g)
    cy = torch.randint(0, H, (B, N, 1), generator=g)
    cx = torch.randint(0, W, (B, N, 1), generator=g)
    c = torch.cat([cy, cx], -1)
    if name == "p3_border":
        c[0, 0] = torch.tensor([0, 0]); c[0, 1] = torch.tensor([H - 1, W - 1]); c[1, 0] = torch.tensor([0, W - 1])
    return t, c.float(), ps


PATCH_CASES = ["p7x5", "p9", "p3_border"]      # odd sizes only: the torch twin reshapes (2*pad+1)^2 windows to h*w (local_ssi.py:69-77)


def _load_file(modname: str, path: str):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def reference_knn_module():
    """The reference's functions/knn.py bound to its compiled CPU extension."""
    assert build_ref_knn.build(), "reference KNN sources not present"
    build_ref_knn.load_ref()
    sys.dont_write_bytecode = True
    return _load_file("_ref_knn_functions", os.path.join(ref_loader.REF_ROOT, "unidepth", "ops", "knn", "functions", "knn.py"))


def reference_chamfer_class(knn_mod):
    """utils/chamfer_distance.py with its `unidepth.ops.knn` import satisfied by the module above."""
    import types
    pkg = types.ModuleType("unidepth.ops.knn")
    pkg.knn_points = knn_mod.knn_points
    pkg.knn_gather = knn_mod.knn_gather
    saved = {k: sys.modules.get(k) for k in ("unidepth", "unidepth.ops", "unidepth.ops.knn")}
    sys.modules.setdefault("unidepth", types.ModuleType("unidepth"))
    sys.modules.setdefault("unidepth.ops", types.ModuleType("unidepth.ops"))
    sys.modules["unidepth.ops.knn"] = pkg
    try:
        mod = _load_file("_ref_chamfer", os.path.join(ref_loader.REF_ROOT, "unidepth", "utils", "chamfer_distance.py"))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod.ChamferDistance


def eval3d_case_inputs(name: str):
    """(gts, preds, masks, thresholds) of the 3-D metric golden cases: 'small' needs no resampling, 'resampled' has more than
    240 * 320 valid points, so eval_3d shrinks the maps with nearest-exact first (evaluation_depth.py:164-172)."""
    seed, B, H, W, keep = {"small": (31, 2, 48, 64, 0.8), "resampled": (32, 2, 260, 340, 0.7)}[name]
    g = torch.Generator().manual_seed(seed)
    z = 1.0 + 4.0 * torch.rand(B, 1, H, W, generator=g)
    yy, xx = torch.meshgrid(torch.linspace(-0.6, 0.6, H), torch.linspace(-0.8, 0.8, W), indexing="ij")
    gts = torch.cat([xx[None, None] * z, yy[None, None] * z, z], 1)
    preds = gts * (1.0 + 0.03 * torch.randn(B, 3, H, W, generator=g)) + 0.01 * torch.randn(B, 3, H, W, generator=g)
    masks = torch.rand(B, 1, H, W, generator=g) < keep
    thresholds = [0.0025, 0.01, 0.04, 0.16]
    return gts, preds, masks, thresholds


EVAL3D_CASES = ["small", "resampled"]


def reference_eval_3d(knn_mod):
    """utils/evaluation_depth.py with its chamfer import satisfied by the reference chamfer over the compiled reference K-NN."""
    import types
    cham_cls = reference_chamfer_class(knn_mod)
    mod = types.ModuleType("unidepth.utils.chamfer_distance")
    mod.ChamferDistance = cham_cls
    saved = {k: sys.modules.get(k) for k in ("unidepth", "unidepth.utils", "unidepth.utils.chamfer_distance")}
    sys.modules.setdefault("unidepth", types.ModuleType("unidepth"))
    sys.modules.setdefault("unidepth.utils", types.ModuleType("unidepth.utils"))
    sys.modules["unidepth.utils.chamfer_distance"] = mod
    try:
        ev = _load_file("_ref_evaluation_depth", os.path.join(ref_loader.REF_ROOT, "unidepth", "utils", "evaluation_depth.py"))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ev.eval_3d


def reference_extract_patches():
    ref_loader._prepare()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from unidepth.ops.losses.local_ssi import extract_patches  # type: ignore
    return extract_patches


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    knn = reference_knn_module()
    out = {}
    for name in KNN_CASES:
        p1, p2, l1, l2, norm, K = knn_case_inputs(name)
        r = knn.knn_points(p1, p2, lengths1=l1, lengths2=l2, norm=norm, K=K, return_nn=True)
        out[f"{name}.dists"] = r.dists.numpy()
        out[f"{name}.idx"] = r.idx.numpy()
        out[f"{name}.knn"] = r.knn.numpy()
    cham = reference_chamfer_class(knn)()
    p1, p2, l1, l2, _, _ = knn_case_inputs("d3_k5_ragged")
    cx, cy, ix, iy = cham(p1, p2, x_lengths=l1, y_lengths=l2)
    out["chamfer.cx"], out["chamfer.cy"], out["chamfer.ix"], out["chamfer.iy"] = cx.numpy(), cy.numpy(), ix.numpy(), iy.numpy()
    ev3d = reference_eval_3d(knn)
    for name in EVAL3D_CASES:
        gts, preds, masks, th = eval3d_case_inputs(name)
        for k, v in ev3d(gts, preds, masks, th).items():
            out[f"eval3d.{name}.{k}"] = v.numpy()
    np.savez_compressed(os.path.join(GOLDEN, "eval_knn.npz"), **out)
    ep = reference_extract_patches()
    out = {}
    for name in PATCH_CASES:
        t, c, ps = patch_case_inputs(name)
        out[name] = ep(t, c, ps).numpy()            # [B, N*C, h*w]
    np.savez_compressed(os.path.join(GOLDEN, "eval_patches.npz"), **out)
    print("wrote", os.path.join(GOLDEN, "eval_knn.npz"), os.path.join(GOLDEN, "eval_patches.npz"))


if __name__ == "__main__":
    main()
