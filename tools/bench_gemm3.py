#!/usr/bin/env python
"""GEMM micro-benchmark with COLD operands: cycles through 24 distinct weight/activation buffers (like 24 layers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidepth_amd import ops
M = 8 * 1376
g = torch.Generator().manual_seed(0)
for name, N, K, kind in [("qkv", 3072, 1024, "qkv"), ("proj", 1024, 1024, "acc"), ("fc1", 4096, 1024, "gelu"), ("fc2", 1024, 4096, "acc")]:
    for hint in (1, 2, 3):
        for mode in ("hot", "coldW", "coldAW"):
            nbuf = 1 if mode == "hot" else 24
            Ws = [(torch.randn(N, K, generator=g) * K ** -0.5).half().cuda() for _ in range(nbuf)]
            As = [torch.randn(M, K, generator=g).half().cuda() for _ in range(nbuf if mode == "coldAW" else 1)]
            bias = torch.randn(N, generator=g).cuda()
            P = ops.Program()
            for i in range(24):
                W = Ws[i % nbuf]; A = As[i % len(As)]
                if kind == "qkv":
                    out = torch.zeros(M, 2048, dtype=torch.half, device="cuda"); vt = torch.zeros(8, 16, 64, 1408, dtype=torch.half, device="cuda")
                    kw = dict(out=out, out2=vt, ldc=2048, epi=ops.UD_EPI_QKV, vsplit=2048, tok_per_img=1376, kv_ld=1408, heads_v=16)
                elif kind == "gelu":
                    out = torch.zeros(M, N, dtype=torch.half, device="cuda"); kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F16, act=ops.UD_ACT_GELU)
                else:
                    out = torch.zeros(M, N, device="cuda"); kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F32, accumulate=1)
                P.gemm(A=A, W=W, bias=bias, M=M, N=N, K=K, lda=K, ldw=K, tile_hint=hint, **kw)
            for _ in range(2): P.run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): P.run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / (5 * 24) * 1e3
            print(f"{name:5s} hint={hint} {mode:7s}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF")
            del Ws, As, P
