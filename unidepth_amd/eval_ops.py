"""Host-side mirror of the reference's two native extensions (evaluation / loss side), over the C-ABI of include/unidepth_hip.h:

    knn_points, knn_gather          unidepth/ops/knn/functions/knn.py:113-196, 199-249  (forward; the KNN extension)
    ChamferDistance, chamfer_dist   unidepth/utils/chamfer_distance.py:60-159, unidepth/utils/evaluation_depth.py:12-18
    auc, f1_score, DICT_METRICS_3D, eval_3d   unidepth/utils/evaluation_depth.py:21-34,74-91,109-125,160-182 (the 3-D metrics)
    RandomPatchExtractor            unidepth/ops/extract_patches/modules/patch_extractor.py:10-42 (forward)

Same names, argument meaning and error behaviour; inference / evaluation only (no autograd: the reference's backward kernels are
training code).  Tensors must live on the GPU: there is no CPU path."""
from __future__ import annotations

from collections import namedtuple
from typing import Optional, Tuple, Union

import torch

from . import _lib
from .ops import check, cur_stream, mk

_KNN = namedtuple("KNN", "dists idx knn")


def _lengths(lengths: Optional[torch.Tensor], N: int, device) -> Optional[torch.Tensor]:
    if lengths is None:
        return None
    if lengths.ndim != 1 or lengths.shape[0] != N:
        raise ValueError("Expected lengths to be of shape (N,)")
    return lengths.to(device=device, dtype=torch.int64).contiguous()


def knn_points(p1: torch.Tensor, p2: torch.Tensor, lengths1: Union[torch.Tensor, None] = None, lengths2: Union[torch.Tensor, None] = None,
               norm: int = 2, K: int = 1, version: int = -1, return_nn: bool = False, return_sorted: bool = True) -> _KNN:
    """K nearest neighbours in p2 of every point of p1 (functions/knn.py:113-196).  `version` selected a CUDA kernel variant in the
    reference and is accepted and ignored; the result is always sorted ascending (what return_sorted=True gives the reference; an
    unsorted result is any order, so this satisfies return_sorted=False too).  dists are squared L2 (norm 2) or L1 (norm 1)."""
    if p1.shape[0] != p2.shape[0]:
        raise ValueError("pts1 and pts2 must have the same batch dimension.")
    if p1.shape[2] != p2.shape[2]:
        raise ValueError("pts1 and pts2 must have the same point dimension.")
    if norm not in (1, 2):
        raise ValueError("Support for 1 or 2 norm.")
    if not p1.is_cuda or not p2.is_cuda:
        raise RuntimeError("knn_points: GPU tensors expected (the HIP kernel is the only implementation)")
    p1 = p1.float().contiguous()
    p2 = p2.float().contiguous()
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    l1 = _lengths(lengths1, N, p1.device)
    l2 = _lengths(lengths2, N, p1.device)
    dists = torch.empty(N, P1, K, device=p1.device, dtype=torch.float32)
    idx = torch.empty(N, P1, K, device=p1.device, dtype=torch.int64)
    if N * P1 * K:
        d = mk(_lib.UdKnn, p1=p1, p2=p2, lengths1=l1, lengths2=l2, dists=dists, idx=idx, work=None, N=N, P1=P1, P2=P2, D=D, K=K, norm=norm)
        if K == 1 and -(-P1 // 512) * N < 1024 and P2 > 4096:
            work = torch.empty(N * P1, device=p1.device, dtype=torch.int64)     # scratch for the P2-split merge
            d.work = work.data_ptr()
        check(_lib.lib.ud_knn_points(d, cur_stream()), "ud_knn_points")
    nn = knn_gather(p2, idx, lengths2) if return_nn else None
    return _KNN(dists=dists, idx=idx, knn=nn)


def knn_gather(x: torch.Tensor, idx: torch.Tensor, lengths: Union[torch.Tensor, None] = None) -> torch.Tensor:
    """x_out[n, l, k] = x[n, idx[n, l, k]] for features x [N, M, U] and neighbour indices idx [N, L, K]; slots k >= lengths[n] (clouds
    with fewer than K points) are zero (functions/knn.py:199-249).  Plain indexing: no kernel of ours."""
    if x.shape[0] != idx.shape[0]:
        raise ValueError("x and idx must have same batch dimension.")
    N, K = idx.shape[0], idx.shape[2]
    batch = torch.arange(N, device=x.device).view(N, 1, 1)
    x_out = x[batch, idx]                                              # [N, L, K, U]
    if lengths is not None:
        dead = torch.arange(K, device=x.device).view(1, 1, K) >= lengths.to(x.device).view(N, 1, 1)
        x_out = x_out.masked_fill(dead.unsqueeze(-1), 0.0)
    return x_out


def _check_cloud(points: torch.Tensor, lengths: Optional[torch.Tensor], normals: Optional[torch.Tensor]) -> torch.Tensor:
    """Shape checks of one side of the Chamfer inputs (chamfer_distance.py:33-57); returns the lengths (full clouds when None)."""
    if points.ndim != 3:
        raise ValueError("Expected points to be of shape (N, P, D)")
    if lengths is None:
        lengths = torch.full((points.shape[0],), points.shape[1], dtype=torch.int64, device=points.device)
    elif lengths.ndim != 1 or lengths.shape[0] != points.shape[0]:
        raise ValueError("Expected lengths to be of shape (N,)")
    if normals is not None and normals.ndim != 3:
        raise ValueError("Expected normals to be of shape (N, P, 3")
    return lengths


class ChamferDistance(torch.nn.Module):
    """Per-point squared distances to the nearest neighbour in the other cloud, both directions (utils/chamfer_distance.py:60-159):
    returns (cham_x [N,P1], cham_y [N,P2], idx_x [N,P1], idx_y [N,P2]).  The reduction arguments are validated and, as in the
    reference, otherwise unused; normals are accepted and ignored (the reference never reads them either)."""

    def forward(self, x, y, x_lengths=None, y_lengths=None, x_normals=None, y_normals=None, weights=None,
                batch_reduction: Union[str, None] = "mean", point_reduction: str = "mean"):
        if batch_reduction is not None and batch_reduction not in ["mean", "sum"]:
            raise ValueError('batch_reduction must be one of ["mean", "sum"] or None')
        if point_reduction not in ["mean", "sum"]:
            raise ValueError('point_reduction must be one of ["mean", "sum"]')
        x_lengths = _check_cloud(x, x_lengths, x_normals)
        y_lengths = _check_cloud(y, y_lengths, y_normals)
        N = x.shape[0]
        if y.shape[0] != N or y.shape[2] != x.shape[2]:
            raise ValueError("y does not have the correct shape.")
        if weights is not None:
            if weights.size(0) != N:
                raise ValueError("weights must be of shape (N,).")
            if bool((weights < 0).any()):
                raise ValueError("weights cannot be negative.")
            if float(weights.sum()) == 0.0:                      # all-zero weights: the reference returns zeros shaped by the reduction
                zero = (x.sum((1, 2)) * weights.view(N)) * 0.0
                return (zero.sum(), zero.sum()) if batch_reduction in ("mean", "sum") else (zero.view(N, 1), zero.view(N, 1))
        x_nn = knn_points(x, y, lengths1=x_lengths, lengths2=y_lengths, K=1)       # rows beyond a length are already zero
        y_nn = knn_points(y, x, lengths1=y_lengths, lengths2=x_lengths, K=1)
        cham_x = x_nn.dists[..., 0]
        cham_y = y_nn.dists[..., 0]
        if weights is not None:
            cham_x = cham_x * weights.view(N, 1)
            cham_y = cham_y * weights.view(N, 1)
        return cham_x, cham_y, x_nn.idx[..., -1], y_nn.idx[..., -1]


def chamfer_dist(tensor1: torch.Tensor, tensor2: torch.Tensor) -> torch.Tensor:
    """(sqrt(d(x->y)) + sqrt(d(y->x))) / 2 per point, clouds of equal size (utils/evaluation_depth.py:12-18)."""
    d1, d2, _, _ = ChamferDistance()(tensor1, tensor2)
    return (torch.sqrt(d1) + torch.sqrt(d2)) / 2


def _precision_recall(tensor1, tensor2, thresholds):
    d1, d2, _, _ = ChamferDistance()(tensor1, tensor2)
    precisions = torch.stack([(d1 < th).sum() / d1.numel() for th in thresholds]).to(tensor1.device)
    recalls = torch.stack([(d2 < th).sum() / d2.numel() for th in thresholds]).to(tensor1.device)
    return precisions, recalls


def auc(tensor1: torch.Tensor, tensor2: torch.Tensor, thresholds) -> torch.Tensor:
    """Area under the precision(recall) curve over the distance thresholds (utils/evaluation_depth.py:21-34).  As in the reference the
    thresholds are compared with SQUARED distances (what knn_points returns)."""
    precisions, recalls = _precision_recall(tensor1, tensor2, thresholds)
    return torch.trapz(precisions, recalls)


def f1_score(tensor1: torch.Tensor, tensor2: torch.Tensor, thresholds) -> torch.Tensor:
    """Mean over thresholds of F1 = 2 P R / (P + R) (0 where undefined), trapezoid rule (utils/evaluation_depth.py:74-91)."""
    precisions, recalls = _precision_recall(tensor1, tensor2, thresholds)
    f1 = 2 * precisions * recalls / (precisions + recalls)
    f1 = torch.where(torch.isnan(f1), torch.zeros_like(f1), f1)
    return torch.trapz(f1) / len(thresholds)


DICT_METRICS_3D = {                  # utils/evaluation_depth.py:109-125; gt / pred: [3, P] point sets of one image
    "MSE_3d": lambda gt, pred, thresholds: torch.norm(gt - pred, dim=0, p=2),
    "chamfer": lambda gt, pred, thresholds: chamfer_dist(gt.unsqueeze(0).permute(0, 2, 1), pred.unsqueeze(0).permute(0, 2, 1)),
    "F1": lambda gt, pred, thresholds: f1_score(gt.unsqueeze(0).permute(0, 2, 1), pred.unsqueeze(0).permute(0, 2, 1), thresholds=thresholds),
}


def eval_3d(gts: torch.Tensor, preds: torch.Tensor, masks: torch.Tensor, thresholds=None):
    """3-D metrics of a batch of point maps [B,3,H,W] under validity masks [B,1,H,W] (utils/evaluation_depth.py:160-182): maps are
    nearest-exact resampled so that about 240 x 320 valid points remain per batch, then per image MSE_3d, chamfer and F1 on the
    masked points.  The nearest-neighbour searches run in ud_knn_points."""
    import torch.nn.functional as F
    from collections import defaultdict
    summary = defaultdict(list)
    ratio = min(1.0, float((240 * 320 / masks.sum()) ** 0.5))
    h_max, w_max = int(gts.shape[-2] * ratio), int(gts.shape[-1] * ratio)
    gts = F.interpolate(gts, size=(h_max, w_max), mode="nearest-exact")
    preds = F.interpolate(preds, size=(h_max, w_max), mode="nearest-exact")
    masks = F.interpolate(masks.float(), size=(h_max, w_max), mode="nearest-exact").bool()
    for gt, pred, mask in zip(gts, preds, masks):
        if not torch.any(mask):
            continue
        for name, fn in DICT_METRICS_3D.items():
            summary[name].append(fn(gt[:, mask.squeeze()], pred[:, mask.squeeze()], thresholds).mean())
    return {name: torch.stack(vals, dim=0) for name, vals in summary.items()}


class RandomPatchExtractor(torch.nn.Module):
    """patch_size = (width, height) patches around `centers` [B,N,2] = (y, x), zero beyond the image border
    (modules/patch_extractor.py:16-42).  The result has the reference's shape {B, C, N, h, w} over memory written in [b][n][c][i][j]
    order (extract_patches_kernel.cu:22 vs :91) -- identical for the C = 1 tensors every reference call site passes."""

    def forward(self, tensor: torch.Tensor, centers: torch.Tensor, patch_size: Tuple[int, int]) -> torch.Tensor:
        if not tensor.is_cuda:
            raise RuntimeError("RandomPatchExtractor: GPU tensors expected (the HIP kernel is the only implementation)")
        dtype = tensor.dtype
        patch_width, patch_height = patch_size
        pad_w, pad_h = patch_width // 2, patch_height // 2
        B, Cc, H, W = tensor.shape
        N = centers.shape[1]
        # the reference shifts the centres into padded coordinates in the image dtype, then truncates (.int()) -- kept, so that
        # fractional / negative centres round the same way
        cpad = (centers.to(tensor.device) + torch.tensor([pad_h, pad_w], dtype=dtype, device=tensor.device).reshape(1, 1, 2)).int().contiguous()
        x = tensor.float().contiguous()
        out = torch.empty(B, Cc, N, patch_height, patch_width, device=tensor.device, dtype=torch.float32)
        d = mk(_lib.UdExtractPatches, in_=x, out=out, centers=cpad, B=B, C=Cc, H=H, W=W, N=N, h=patch_height, w=patch_width,
               pad_h=pad_h, pad_w=pad_w)
        check(_lib.lib.ud_extract_patches(d, cur_stream()), "ud_extract_patches")
        return out.to(dtype)
