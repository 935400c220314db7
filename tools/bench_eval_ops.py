"""Times the evaluation-side kernels on the GPU box: K-NN / Chamfer at the size the 3-D metrics run on (one 480 x 640 depth map per
cloud, D = 3, K = 1, both directions) and the patch gather; the reference's own CPU K-NN (oracle/_ref/knn/KNN.so, compiled from its
sources) is timed beside it on a bounded sample of the same queries.  Prints one JSON line.   python tools/bench_eval_ops.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidepth_amd import eval_ops  # noqa: E402


def gpu_ms(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))[reps // 2]


def main():
    g = torch.Generator().manual_seed(0)
    P = 480 * 640
    x = torch.randn(1, P, 3, generator=g)
    y = x[:, torch.randperm(P, generator=g)] + 0.01 * torch.randn(1, P, 3, generator=g)
    xd, yd = x.cuda(), y.cuda()
    out = {"op": "knn_points D=3 K=1", "P1": P, "P2": P}
    ms = gpu_ms(lambda: eval_ops.knn_points(xd, yd, K=1))
    out["knn_ms"] = round(ms, 3)
    out["pairs_per_s"] = round(P * P / ms * 1e3, 0)
    out["valu_tflops"] = round(P * P * 8 / ms * 1e3 / 1e12, 2)          # 3 sub + 3 mul + 2 add per pair (compare/select not counted)
    # the kernel is bound by the fp32 vector pipe; its arithmetic may not be fused (bit-exact distances), so the ceiling is the packed
    # NON-FMA rate: half of the 157.3 TFLOP/s packed-FMA vector peak (MI355X_MICROARCH.md)
    out["roofline"] = {"bound": "valu_f32", "achieved": out["valu_tflops"], "peak": 78.6, "unit": "TFLOP/s (fp32 add/mul, no FMA)",
                       "frac": round(out["valu_tflops"] / 78.6, 3), "flop_per_launch": 8.0 * P * P, "avg_launch_us": round(ms * 1e3, 1)}
    out["chamfer_ms"] = round(gpu_ms(lambda: eval_ops.chamfer_dist(xd, yd)), 3)
    for K in (4, 8):
        out[f"knn_k{K}_ms"] = round(gpu_ms(lambda: eval_ops.knn_points(xd, yd, K=K), 3), 3)
    xs, ys = xd[:, :4096].contiguous(), yd
    out["knn_small_p1_split_ms"] = round(gpu_ms(lambda: eval_ops.knn_points(xs, ys, K=1)), 3)
    img = torch.randn(8, 1, 480, 640, device="cuda")
    cen = torch.stack([torch.randint(0, 480, (8, 4096), device="cuda"), torch.randint(0, 640, (8, 4096), device="cuda")], -1).float()
    ext = eval_ops.RandomPatchExtractor()
    ms = gpu_ms(lambda: ext(img, cen, (32, 32)))
    out["extract_patches_ms"] = round(ms, 3)
    out["extract_patches_GBps"] = round(8 * 4096 * 32 * 32 * 4 * 2 / ms / 1e6, 1)   # read + write of every patch element
    from oracle import build_ref_knn                                                 # CPU baseline leg only
    ref = build_ref_knn.load_ref()
    if ref is not None:
        nq = 16384
        l1, l2 = torch.tensor([nq]), torch.tensor([P])
        torch.set_num_threads(1)
        t0 = time.perf_counter()
        idx, d = ref.knn_points_idx(x[:, :nq].contiguous(), y, l1, l2, 2, 1, -1)
        dt = time.perf_counter() - t0
        r = eval_ops.knn_points(xd[:, :nq].contiguous(), yd, K=1)
        out["cpu_reference"] = {"kind": "reference", "sample": f"{nq} of {P} queries against all {P} points", "cores": 1,
                                "pairs_per_s": round(nq * P / dt, 0), "seconds": round(dt, 2),
                                "matches_gpu": bool(torch.equal(r.idx.cpu(), idx) and torch.equal(r.dists.cpu(), d))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
