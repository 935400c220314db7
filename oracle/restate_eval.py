"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the reference's two native extensions, forward semantics.

Only tests/, __graft_entry__.smoke() and benchmark baselines may import this module; the product path (unidepth_amd/eval_ops.py ->
libunidepth_hip.so) never does.

PINNED: knn_points against the REFERENCE's own CPU implementation compiled from its sources (oracle/build_ref_knn.py ->
oracle/_ref/knn/KNN.so, driven through the reference's Python wrapper unidepth/ops/knn/functions/knn.py) and extract_patches against the
reference's pure-torch twin of its CUDA kernel (unidepth/ops/losses/local_ssi.py:44-77); tests/test_oracle_eval_pins.py runs both
live in the authoring container and against the golden vectors tests/golden/eval_*.npz (written by oracle/make_golden_eval.py from
those reference runs) everywhere.  The reference's CUDA kernels themselves cannot run here (no CUDA): for K > 1 with exactly tied
distances its GPU MinK (utils/mink.cuh:55-72) may keep a different tied neighbour than its CPU priority queue; this restatement and
the HIP kernel follow the CPU definition ((dist, index) lexicographic), which is also the GPU result whenever there are no ties."""
from __future__ import annotations

import numpy as np


def knn_points(p1: np.ndarray, p2: np.ndarray, lengths1=None, lengths2=None, norm: int = 2, K: int = 1):
    """(dists [N,P1,K] f32, idx [N,P1,K] i64): unidepth/ops/knn/src/knn_cpu.cpp:13-70 followed by the sort of
    functions/knn.py:75-91.  dist is accumulated d = 0..D-1 in fp32 with separately rounded multiply and add (knn_cpu.cpp:43-50);
    the queue keeps the K smallest (dist, index) tuples (:51-58) and empties largest-first into slots K-1..0 (:60-66), so the
    output is ascending and tie-broken by index; unfilled slots / rows beyond lengths1 stay 0 (:26-27)."""
    p1 = np.ascontiguousarray(p1, np.float32)
    p2 = np.ascontiguousarray(p2, np.float32)
    if p1.shape[0] != p2.shape[0]:
        raise ValueError("pts1 and pts2 must have the same batch dimension.")
    if p1.shape[2] != p2.shape[2]:
        raise ValueError("pts1 and pts2 must have the same point dimension.")
    if norm not in (1, 2):
        raise ValueError("Support for 1 or 2 norm.")
    N, P1, D = p1.shape
    P2 = p2.shape[1]
    dists = np.zeros((N, P1, K), np.float32)
    idx = np.zeros((N, P1, K), np.int64)
    for n in range(N):
        l1 = P1 if lengths1 is None else int(lengths1[n])
        l2 = P2 if lengths2 is None else int(lengths2[n])
        if l1 == 0 or l2 == 0:
            continue
        a, b = p1[n, :l1], p2[n, :l2]
        for s in range(0, l1, 2048):                                  # chunked [chunk, l2] distance matrix
            q = a[s:s + 2048]
            dist = np.zeros((q.shape[0], l2), np.float32)
            for d in range(D):
                diff = q[:, d:d + 1] - b[None, :, d]
                dist = dist + (diff * diff if norm == 2 else np.abs(diff))       # two roundings, like the scalar C loop
            if K == 1:
                order = np.argmin(dist, axis=1)[:, None]                 # first minimum = lowest index among ties
            else:
                order = np.argsort(dist, axis=1, kind="stable")[:, :K]   # stable: ties -> lower index first
            kk = order.shape[1]
            dists[n, s:s + q.shape[0], :kk] = np.take_along_axis(dist, order, 1)
            idx[n, s:s + q.shape[0], :kk] = order
    return dists, idx


def knn_gather(x: np.ndarray, idx: np.ndarray, lengths=None) -> np.ndarray:
    """functions/knn.py:199-249."""
    N, M, U = x.shape
    K = idx.shape[2]
    out = np.stack([x[n][idx[n]] for n in range(N)])
    if lengths is not None:
        for n in range(N):
            if lengths[n] < K:
                out[n, :, int(lengths[n]):] = 0.0
    return out


def chamfer(x: np.ndarray, y: np.ndarray, x_lengths=None, y_lengths=None):
    """utils/chamfer_distance.py:143-159 (weights None): nearest-neighbour squared distances both ways, masked rows zero."""
    dx, ix = knn_points(x, y, x_lengths, y_lengths, K=1)
    dy, iy = knn_points(y, x, y_lengths, x_lengths, K=1)
    return dx[..., 0], dy[..., 0], ix[..., -1], iy[..., -1]


def chamfer_dist(t1: np.ndarray, t2: np.ndarray) -> np.ndarray:
    """utils/evaluation_depth.py:12-18."""
    d1, d2, _, _ = chamfer(t1, t2)
    return (np.sqrt(d1) + np.sqrt(d2)) / 2


def extract_patches(tensor: np.ndarray, centers: np.ndarray, patch_size) -> np.ndarray:
    """RandomPatchExtractor.forward (modules/patch_extractor.py:16-42) over extract_patches_cuda_forward_kernel
    (extract_patches_kernel.cu:65-95): zero-pad by half a patch, shift the centres (in the image dtype, then truncate), copy
    h x w windows; memory order [b][n][c][i][j] presented with shape (B, C, N, h, w) (:22 vs :91)."""
    B, C, H, W = tensor.shape
    pw, ph = patch_size
    pad_w, pad_h = pw // 2, ph // 2
    padded = np.zeros((B, C, H + 2 * pad_h, W + 2 * pad_w), np.float32)
    padded[:, :, pad_h:pad_h + H, pad_w:pad_w + W] = tensor.astype(np.float32)
    c = np.trunc(centers.astype(tensor.dtype if tensor.dtype.kind == "f" else np.float32) + np.array([pad_h, pad_w], np.float32)).astype(np.int64)
    N = c.shape[1]
    out = np.zeros((B, N, C, ph, pw), np.float32)
    for b in range(B):
        for n in range(N):
            y0, x0 = c[b, n, 0] - ph // 2, c[b, n, 1] - pw // 2
            for i in range(ph):
                for j in range(pw):
                    y, x = y0 + i, x0 + j
                    if 0 <= y < padded.shape[2] and 0 <= x < padded.shape[3]:
                        out[b, n, :, i, j] = padded[b, :, y, x]
    return out.reshape(B, C, N, ph, pw).astype(tensor.dtype if tensor.dtype.kind == "f" else np.float32)


def f1_score(t1: np.ndarray, t2: np.ndarray, thresholds) -> np.float32:
    """utils/evaluation_depth.py:74-91 (thresholds against SQUARED distances, as the reference compares them)."""
    d1, d2, _, _ = chamfer(t1, t2)
    pr = np.array([np.float32((d1 < th).sum()) / np.float32(d1.size) for th in thresholds], np.float32)
    rc = np.array([np.float32((d2 < th).sum()) / np.float32(d2.size) for th in thresholds], np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        f1 = 2 * pr * rc / (pr + rc)
    f1 = np.where(np.isnan(f1), np.float32(0), f1).astype(np.float32)
    return np.float32(np.trapezoid(f1.astype(np.float64)) / len(thresholds))


def eval_3d(gts, preds, masks, thresholds):
    """utils/evaluation_depth.py:160-182 for inputs that need no resampling (masks.sum() <= 240 * 320): per image MSE_3d, chamfer
    and F1 over the masked points, each reduced by .mean().  gts / preds [B,3,H,W] float32, masks [B,1,H,W] bool."""
    assert masks.sum() <= 240 * 320, "resampling branch: covered by the golden vectors only"
    out = {"MSE_3d": [], "chamfer": [], "F1": []}
    for gt, pred, mask in zip(gts, preds, masks):
        if not mask.any():
            continue
        m = mask[0]
        g, p = gt[:, m], pred[:, m]                                     # [3, P]
        out["MSE_3d"].append(np.sqrt(((g - p).astype(np.float32) ** 2).sum(0)).mean(dtype=np.float32))
        out["chamfer"].append(chamfer_dist(g.T[None], p.T[None]).mean(dtype=np.float32))
        out["F1"].append(f1_score(g.T[None], p.T[None], thresholds))
    return {k: np.array(v, np.float32) for k, v in out.items()}
