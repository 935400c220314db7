"""GPU parity of the two evaluation-side kernels (unidepth_amd/eval_ops.py -> ud_knn_points / ud_extract_patches) against the
reference-derived golden vectors, the CPU restatement (oracle/restate_eval.py) and, when it travelled with the snapshot, the
reference's own CPU K-NN compiled from its sources (oracle/_ref/knn/KNN.so).  Integer / index work: bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import build_ref_knn, make_golden_eval as mg, restate_eval as re_

pytestmark = pytest.mark.gpu


def _dev(t):
    return None if t is None else t.cuda()


@pytest.mark.parametrize("name", mg.KNN_CASES)
def test_knn_points_matches_reference_golden(golden_dir, name):
    from unidepth_amd import eval_ops
    g = np.load(os.path.join(golden_dir, "eval_knn.npz"))
    p1, p2, l1, l2, norm, K = mg.knn_case_inputs(name)
    r = eval_ops.knn_points(p1.cuda(), p2.cuda(), _dev(l1), _dev(l2), norm=norm, K=K, return_nn=True)
    assert np.array_equal(r.dists.cpu().numpy(), g[f"{name}.dists"])
    assert np.array_equal(r.idx.cpu().numpy(), g[f"{name}.idx"])
    assert np.array_equal(r.knn.cpu().numpy(), g[f"{name}.knn"])


def test_chamfer_matches_reference_golden(golden_dir):
    from unidepth_amd import eval_ops
    g = np.load(os.path.join(golden_dir, "eval_knn.npz"))
    p1, p2, l1, l2, _, _ = mg.knn_case_inputs("d3_k5_ragged")
    out = eval_ops.ChamferDistance()(p1.cuda(), p2.cuda(), x_lengths=l1.cuda(), y_lengths=l2.cuda())
    for a, k in zip(out, ("cx", "cy", "ix", "iy")):
        assert np.array_equal(a.cpu().numpy(), g[f"chamfer.{k}"]), k


@pytest.mark.parametrize("shape", [(1, 5000, 7000, 3, 1, 2), (2, 300, 20000, 3, 1, 2), (1, 1000, 3000, 3, 8, 2), (2, 257, 1025, 4, 2, 1),
                                   (1, 500, 900, 7, 16, 2), (1, 300, 600, 20, 3, 2), (1, 100, 5000, 3, 32, 2),
                                   (1, 200, 300, 32, 2, 2), (2, 100, 1000, 1, 1, 2), (1, 64, 700, 9, 1, 1), (3, 1, 1, 3, 1, 2)])
def test_knn_points_matches_restatement_at_larger_sizes(shape):
    """Includes the K = 1 path that splits P2 across blocks (small P1, large P2) and merges through the packed atomic."""
    from unidepth_amd import eval_ops
    N, P1, P2, D, K, norm = shape
    g = torch.Generator().manual_seed(sum(shape))
    p1, p2 = torch.randn(N, P1, D, generator=g), torch.randn(N, P2, D, generator=g)
    l1 = torch.randint(P1 // 2, P1 + 1, (N,), generator=g)
    l2 = torch.randint(P2 // 2, P2 + 1, (N,), generator=g)
    r = eval_ops.knn_points(p1.cuda(), p2.cuda(), l1.cuda(), l2.cuda(), norm=norm, K=K)
    d, i = re_.knn_points(p1.numpy(), p2.numpy(), l1.numpy(), l2.numpy(), norm, K)
    assert np.array_equal(r.dists.cpu().numpy(), d)
    assert np.array_equal(r.idx.cpu().numpy(), i)


def test_knn_points_matches_compiled_reference_when_present():
    ref = build_ref_knn.load_ref()
    if ref is None:
        pytest.skip("oracle/_ref/knn/KNN.so not in this snapshot")
    from unidepth_amd import eval_ops
    g = torch.Generator().manual_seed(5)
    p1, p2 = torch.randn(2, 3000, 3, generator=g), torch.randn(2, 4000, 3, generator=g)
    l1, l2 = torch.tensor([3000, 1234]), torch.tensor([4000, 17])
    for K in (1, 5):
        idx, dists = ref.knn_points_idx(p1, p2, l1, l2, 2, K, -1)
        r = eval_ops.knn_points(p1.cuda(), p2.cuda(), l1.cuda(), l2.cuda(), K=K)
        assert np.array_equal(r.idx.cpu().numpy(), idx.numpy())
        assert np.array_equal(r.dists.cpu().numpy(), dists.numpy())


def test_knn_properties_at_evaluation_size():
    """Size-independent checks at the size the 3-D metrics run on (one 480 x 640 depth map = 307200 points per cloud):
    self-query returns itself at distance 0; distances recomputed from the returned index agree bit-for-bit; no other point of a
    random probe set is closer."""
    from unidepth_amd import eval_ops
    g = torch.Generator().manual_seed(9)
    P = 480 * 640
    x = torch.randn(1, P, 3, generator=g).cuda()
    y = (x + 0.01 * torch.randn(1, P, 3, generator=g).cuda())[:, torch.randperm(P, generator=g).cuda()]
    r = eval_ops.knn_points(x, x, K=1)
    assert torch.equal(r.idx[0, :, 0], torch.arange(P, device="cuda")) and float(r.dists.abs().max()) == 0.0
    r = eval_ops.knn_points(x, y, K=1)
    nn = y[0, r.idx[0, :, 0]]
    diff = x[0] - nn
    rec = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
    assert torch.equal(rec, r.dists[0, :, 0])
    probe = torch.randint(0, P, (512,), generator=g).cuda()
    dprobe = ((x[0, :2048, None, :] - y[0, probe][None]) ** 2).sum(-1)
    assert bool((dprobe.min(1).values >= r.dists[0, :2048, 0] - 1e-6).all())
    cd = eval_ops.chamfer_dist(x, y)
    assert cd.shape == (1, P) and float(cd.mean()) < 0.05


def test_knn_argument_errors():
    from unidepth_amd import eval_ops
    a = torch.zeros(1, 4, 3, device="cuda")
    with pytest.raises(ValueError):
        eval_ops.knn_points(a, torch.zeros(2, 4, 3, device="cuda"))
    with pytest.raises(ValueError):
        eval_ops.knn_points(a, torch.zeros(1, 4, 2, device="cuda"))
    with pytest.raises(ValueError):
        eval_ops.knn_points(a, a, norm=3)
    with pytest.raises(RuntimeError):
        eval_ops.knn_points(a.cpu(), a.cpu())
    with pytest.raises(RuntimeError):
        eval_ops.knn_points(a, a, K=33)
    r = eval_ops.knn_points(a, torch.zeros(1, 0, 3, device="cuda"), K=2)          # empty p2: all padding
    assert float(r.dists.abs().sum()) == 0 and int(r.idx.abs().sum()) == 0


@pytest.mark.parametrize("name", mg.PATCH_CASES)
def test_patch_extractor_matches_reference_golden(golden_dir, name):
    from unidepth_amd import eval_ops
    g = np.load(os.path.join(golden_dir, "eval_patches.npz"))
    t, c, ps = mg.patch_case_inputs(name)
    out = eval_ops.RandomPatchExtractor()(t.cuda(), c.cuda(), ps)
    assert out.shape == (t.shape[0], 1, c.shape[1], ps[1], ps[0])
    assert np.array_equal(out.cpu().numpy().reshape(t.shape[0], -1, ps[0] * ps[1]), g[name])


@pytest.mark.parametrize("cfg", [(2, 3, 40, 50, 13, (4, 2)), (1, 1, 480, 640, 4000, (32, 32)), (2, 2, 9, 9, 5, (11, 15)), (1, 1, 8, 8, 0, (3, 3))])
def test_patch_extractor_matches_restatement(cfg):
    """Even sizes, C > 1 (the [b][n][c] memory order behind the {B,C,N,h,w} shape), patches larger than the image, fp16 input, N = 0."""
    from unidepth_amd import eval_ops
    B, C, H, W, N, ps = cfg
    g = torch.Generator().manual_seed(B * 1000 + N)
    t = torch.randn(B, C, H, W, generator=g)
    c = torch.cat([torch.randint(0, H, (B, N, 1), generator=g), torch.randint(0, W, (B, N, 1), generator=g)], -1).float()
    out = eval_ops.RandomPatchExtractor()(t.cuda(), c.cuda(), ps)
    assert np.array_equal(out.cpu().numpy(), re_.extract_patches(t.numpy(), c.numpy(), ps))
    out16 = eval_ops.RandomPatchExtractor()(t.half().cuda(), c.half().cuda(), ps)
    assert out16.dtype == torch.float16
    if N:
        assert np.array_equal(out16.cpu().numpy(), re_.extract_patches(t.half().numpy(), c.half().numpy(), ps))


@pytest.mark.parametrize("name", mg.EVAL3D_CASES)
def test_eval_3d_matches_reference_golden(golden_dir, name):
    """eval_3d (utils/evaluation_depth.py:160-182) end to end -- nearest-exact resampling to ~240 x 320 valid points, masked point
    sets, chamfer and F1 through ud_knn_points -- against what the reference computed with its own K-NN."""
    from unidepth_amd import eval_ops
    g = np.load(os.path.join(golden_dir, "eval_knn.npz"))
    gts, preds, masks, th = mg.eval3d_case_inputs(name)
    out = eval_ops.eval_3d(gts.cuda(), preds.cuda(), masks.cuda(), th)
    assert set(out) == {"MSE_3d", "chamfer", "F1"}
    for k, v in out.items():
        assert np.allclose(v.cpu().numpy(), g[f"eval3d.{name}.{k}"], rtol=5e-6, atol=0), (k, v, g[f"eval3d.{name}.{k}"])
    p1, p2 = gts[0].reshape(3, -1).t()[None].cuda(), preds[0].reshape(3, -1).t()[None].cuda()
    a = eval_ops.auc(p1, p2, th)                                       # evaluation_depth.py:21-34
    d1, d2 = re_.chamfer(p1.cpu().numpy(), p2.cpu().numpy())[:2] if name == "small" else (None, None)
    if d1 is not None:
        pr = torch.tensor([(d1 < t).sum() / d1.size for t in th], dtype=torch.float32)
        rc = torch.tensor([(d2 < t).sum() / d2.size for t in th], dtype=torch.float32)
        assert abs(float(a) - float(torch.trapz(pr, rc))) < 1e-6
