#!/usr/bin/env python
"""The product's large-tile GEMM K loop on plain problems (fp16 out = A W^T + bias, no fused work): 4096^3 / 8192^3 and the four encoder
shapes, on the SAME operand fill as tools/ubench/gemm8p (uniform [-1, 1) activations, uniform * 0.05 weights), per tile schedule
(tile_hint 2 = 256-row tiles, 3 = 192-row tiles, 8 = row-balanced, 0 = the cost model's pick).  VERDICT r5 item 1(a): is the gap to the
guide's 256^2 template (1320-1340 TFLOP/s at 4096^3 on random operands) in the loop or in the shape?  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidepth_amd import ops

shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (11008, 1024, 1024), (11008, 3072, 1024), (11008, 4096, 1024), (11008, 1024, 4096)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
g = torch.Generator().manual_seed(0)
for M, N, K in shapes:
    A = (torch.rand(M, K, generator=g) * 2 - 1).half().cuda()
    W = ((torch.rand(N, K, generator=g) * 2 - 1) * 0.05).half().cuda()
    bias = torch.zeros(N).cuda()
    ref = None
    row = []
    for hint in (2, 3, 8, 0):
        out = torch.zeros(M, N, dtype=torch.half, device="cuda")
        P = ops.Program()
        try:
            P.gemm(A=A, W=W, bias=bias, M=M, N=N, K=K, lda=K, ldw=K, tile_hint=hint, out=out, ldc=N, epi=ops.UD_EPI_F16)
        except Exception as e:
            row.append(f"hint {hint}: {e}")
            continue
        for _ in range(3):
            P.run()
        torch.cuda.synchronize()
        if ref is None:
            ref = (A[:256].float() @ W.float().t()).half()
            err = float((out[:256].float() - ref.float()).abs().max() / ref.float().abs().max())
        best, tot = 1e30, 0.0
        R = 5
        for r in range(R):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                P.run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 10 * 1e3
            best = min(best, us); tot += us
        us = tot / R
        row.append(f"hint {hint}: {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.0f} TF (best {2.0 * M * N * K / best / 1e6:5.0f})")
    print(f"product M {M} N {N} K {K}: " + " | ".join(row) + f" | rel err {err:.1e}", flush=True)
