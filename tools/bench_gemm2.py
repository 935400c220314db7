#!/usr/bin/env python
"""Sweep K and epilogue for fixed (M, N) to separate per-tile fixed cost from the K-loop rate.  GPU box only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unidepth_amd import ops

M0 = 8 * 1376
g = torch.Generator().manual_seed(0)


def run(N, K, kind, hint, M=M0):
    A = (torch.randn(M, K, generator=g)).half().cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    if kind == "f16":
        out = torch.zeros(M, N, dtype=torch.half, device="cuda"); kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F16)
    elif kind == "gelu":
        out = torch.zeros(M, N, dtype=torch.half, device="cuda"); kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F16, act=ops.UD_ACT_GELU)
    elif kind == "f32":
        out = torch.zeros(M, N, device="cuda"); kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F32)
    else:
        out = torch.zeros(M, N, device="cuda"); kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F32, accumulate=1)
    P = ops.Program()
    P.gemm(A=A, W=W, bias=bias, M=M, N=N, K=K, lda=K, ldw=K, tile_hint=hint, **kw)
    for _ in range(3):
        P.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        P.run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


for hint in (1, 2):
    print("tile_hint", hint)
    for N in (4096, 1024):
        for kind in ("f16", "gelu", "f32", "acc"):
            ts = [run(N, K, kind, hint) for K in (64, 256, 1024, 4096)]
            print(f"  N={N} {kind:5s}: " + "  ".join(f"K={K}: {t:7.1f}us" for K, t in zip((64, 256, 1024, 4096), ts)))
for Mx in (256 * 64, 256 * 128, 256 * 256):
    print("M", Mx, "N=1024 f16 big: K=1024", f"{run(1024, 1024, 'f16', 2, M=Mx):.1f}us", " K=4096:", f"{run(1024, 4096, 'f16', 2, M=Mx):.1f}us")
