"""Pins for oracle/restate_eval.py: the golden vectors written from the reference itself (oracle/make_golden_eval.py: the reference's
Python knn wrapper over its own CPU extension compiled from its sources; its pure-torch patch extractor) and, in the authoring
container, the live reference."""
import os

import numpy as np
import pytest
import torch

from oracle import build_ref_knn, make_golden_eval as mg, ref_loader, restate_eval as re_


@pytest.fixture(scope="module")
def gknn(golden_dir):
    return np.load(os.path.join(golden_dir, "eval_knn.npz"))


@pytest.mark.parametrize("name", mg.KNN_CASES)
def test_knn_restatement_matches_reference_golden(gknn, name):
    p1, p2, l1, l2, norm, K = mg.knn_case_inputs(name)
    d, i = re_.knn_points(p1.numpy(), p2.numpy(), None if l1 is None else l1.numpy(), None if l2 is None else l2.numpy(), norm, K)
    assert np.array_equal(d, gknn[f"{name}.dists"])            # bit-exact fp32 distances
    assert np.array_equal(i, gknn[f"{name}.idx"])              # including the tied lattice case and short clouds
    nn = re_.knn_gather(p2.numpy(), i, None if l2 is None else l2.numpy())
    assert np.array_equal(nn, gknn[f"{name}.knn"])


def test_chamfer_restatement_matches_reference_golden(gknn):
    p1, p2, l1, l2, _, _ = mg.knn_case_inputs("d3_k5_ragged")
    cx, cy, ix, iy = re_.chamfer(p1.numpy(), p2.numpy(), l1.numpy(), l2.numpy())
    for a, k in ((cx, "cx"), (cy, "cy"), (ix, "ix"), (iy, "iy")):
        assert np.array_equal(a, gknn[f"chamfer.{k}"]), k


@pytest.mark.parametrize("name", mg.PATCH_CASES)
def test_patch_restatement_matches_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "eval_patches.npz"))
    t, c, ps = mg.patch_case_inputs(name)
    out = re_.extract_patches(t.numpy(), c.numpy(), ps)
    B = t.shape[0]
    assert np.array_equal(out.reshape(B, -1, ps[0] * ps[1]), g[name])
    assert np.abs(g[name]).sum() > 0


def test_patch_restatement_even_sizes_follow_the_kernel_indexing():
    """Even patch sizes exist only in the CUDA kernel (the torch twin cannot reshape them): rows cy - h/2 .. cy + h/2 - 1
    (extract_patches_kernel.cu:84-92)."""
    t = np.arange(100, dtype=np.float32).reshape(1, 1, 10, 10)
    out = re_.extract_patches(t, np.array([[[4, 4], [0, 9]]], np.float32), (4, 2))      # w = 4, h = 2
    assert out.shape == (1, 1, 2, 2, 4)
    assert np.array_equal(out[0, 0, 0], t[0, 0, 3:5, 2:6])
    # border case spelled out: centre (0, 9), rows -1..0, cols 7..10 -> row -1 and col 10 are padding
    assert np.array_equal(out[0, 0, 1], np.array([[0, 0, 0, 0], [7, 8, 9, 0]], np.float32))


@pytest.mark.skipif(not ref_loader.available(), reason="/root/reference not present (GPU box)")
def test_knn_restatement_matches_live_reference_build():
    knn = mg.reference_knn_module()
    g = torch.Generator().manual_seed(77)
    for (N, P1, P2, D, K, norm) in ((2, 150, 170, 3, 1, 2), (1, 64, 200, 5, 7, 1), (3, 33, 20, 3, 32, 2)):
        p1, p2 = torch.randn(N, P1, D, generator=g), torch.randn(N, P2, D, generator=g)
        l2 = torch.randint(1, P2 + 1, (N,), generator=g)
        r = knn.knn_points(p1, p2, lengths2=l2, norm=norm, K=K)
        d, i = re_.knn_points(p1.numpy(), p2.numpy(), None, l2.numpy(), norm, K)
        assert np.array_equal(d, r.dists.numpy()) and np.array_equal(i, r.idx.numpy())
    assert os.path.exists(build_ref_knn.SO_PATH)


def test_eval3d_restatement_matches_reference_golden(gknn):
    """MSE_3d / chamfer / F1 of utils/evaluation_depth.py:160-182 run by the reference itself (no-resampling case)."""
    gts, preds, masks, th = mg.eval3d_case_inputs("small")
    out = re_.eval_3d(gts.numpy(), preds.numpy(), masks.numpy(), th)
    for k in ("MSE_3d", "chamfer", "F1"):
        assert np.allclose(out[k], gknn[f"eval3d.small.{k}"], rtol=2e-6, atol=0), k
    assert out["F1"].min() > 0.3 and out["F1"].max() < 0.9          # thresholds straddle the error level: the metric is informative
