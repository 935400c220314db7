#!/usr/bin/env python
"""Micro-benchmark of the encoder GEMM shapes (ViT-L/14, bs=8): 128x128 kernel vs 256x256 kernel.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidepth_amd import ops

M = 8 * 1376
shapes = [("qkv", 3072, 1024, "qkv"), ("proj", 1024, 1024, "f32"), ("fc1", 4096, 1024, "f16"), ("fc2", 1024, 4096, "f32"),
          ("adapter", 512, 1024, "f32"), ("dec_fc1", 2048, 512, "f16"), ("dec_fc2", 512, 2048, "f32")]
g = torch.Generator().manual_seed(0)
for name, N, K, kind in shapes:
    A = (torch.randn(M, K, generator=g)).half().cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = []
    for hint in (1, 2, 3):
        if kind == "qkv":
            out = torch.zeros(M, 2048, dtype=torch.half, device="cuda"); vt = torch.zeros(8, 16, 64, 1408, dtype=torch.half, device="cuda")
            kw = dict(out=out, out2=vt, ldc=2048, epi=ops.UD_EPI_QKV, vsplit=2048, tok_per_img=1376, kv_ld=1408, heads_v=16)
        elif kind == "f16":
            out = torch.zeros(M, N, dtype=torch.half, device="cuda")
            kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F16, act=ops.UD_ACT_GELU)
        else:
            out = torch.zeros(M, N, device="cuda")
            kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F32, accumulate=1)
        P = ops.Program()
        P.gemm(A=A, W=W, bias=bias, M=M, N=N, K=K, lda=K, ldw=K, tile_hint=hint, **kw)
        for _ in range(3):
            P.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            P.run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        res.append((us, 2.0 * M * N * K / us / 1e6))
    print(f"{name:8s} N={N:5d} K={K:5d}  128x128: {res[0][0]:7.1f} us {res[0][1]:7.1f} TF | 256x256: {res[1][0]:7.1f} us {res[1][1]:7.1f} TF | 192x256: {res[2][0]:7.1f} us {res[2][1]:7.1f} TF")
