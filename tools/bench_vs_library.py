#!/usr/bin/env python
"""The four encoder GEMM shapes (ViT-L/14, bs=8: M = 11008 token rows) -- this engine's gemm256 kernels WITH their fused epilogues against the
vendor library's plain fp16 GEMM (torch.matmul -> hipBLASLt / rocBLAS, no bias / GELU / scatter / accumulate), interleaved rounds on one box,
same random operands (the shader clock depends on operand entropy: DESIGN 8.3).  The library call is a yardstick for what a tuned
gfx950 GEMM reaches on these shapes, not a code path of the engine.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidepth_amd import ops
M = 8 * 1376
g = torch.Generator().manual_seed(0)
SHAPES = {"qkv": (3072, 1024, "qkv"), "proj": (1024, 1024, "acc"), "fc1": (4096, 1024, "gelu"), "fc2": (1024, 4096, "acc")}
mine, lib = {}, {}
for k, (N, K, kind) in SHAPES.items():
    A = torch.randn(M, K, generator=g).half().cuda(); W = (torch.randn(N, K, generator=g) * K ** -0.5).half().cuda(); bias = torch.randn(N, generator=g).cuda()
    P = ops.Program()
    if kind == "gelu":
        out = torch.zeros(M, N, dtype=torch.half, device="cuda"); kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F16, act=ops.UD_ACT_GELU)
    elif kind == "acc":
        out = torch.zeros(M, N, device="cuda"); kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F32, accumulate=1)
    else:
        D = N // 3
        out = torch.zeros(M, 2 * D, dtype=torch.half, device="cuda"); vt = torch.zeros(8, D // 64, 64, 1408, dtype=torch.half, device="cuda")
        kw = dict(out=out, out2=vt, ldc=2 * D, epi=ops.UD_EPI_QKV, vsplit=2 * D, tok_per_img=1376, kv_ld=1408, heads_v=D // 64)
    P.gemm(A=A, W=W, bias=bias, M=M, N=N, K=K, lda=K, ldw=K, **kw)
    mine[k] = P
    Wt = W.t().contiguous()                     # library: both weight layouts, the better one counts
    o2 = torch.empty(M, N, dtype=torch.half, device="cuda")
    lib[k] = (A, W, Wt, o2)
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
R = 5
acc = {k: [0.0, 0.0, 0.0] for k in SHAPES}
for r in range(R + 1):
    for k in SHAPES:
        A, W, Wt, o2 = lib[k]
        a = t(mine[k].run)
        b = t(lambda: torch.matmul(A, W.t(), out=o2))
        c = t(lambda: torch.matmul(A, Wt, out=o2))
        if r:
            for i, v in enumerate((a, b, c)): acc[k][i] += v / R
print(f"{'shape':5s} {'N':>5s} {'K':>5s} | engine (fused epilogue)  | library NT (W [N][K])  | library NN (W^T [K][N])")
for k, (N, K, _) in SHAPES.items():
    fl = 2.0 * M * N * K
    a, b, c = acc[k]
    print(f"{k:5s} {N:5d} {K:5d} | {a:7.1f} us {fl / a / 1e6:5.0f} TFLOP/s | {b:7.1f} us {fl / b / 1e6:5.0f} TFLOP/s | {c:7.1f} us {fl / c / 1e6:5.0f} TFLOP/s")
print("engine sum %.1f us; library best-layout sum %.1f us" % (sum(v[0] for v in acc.values()), sum(min(v[1], v[2]) for v in acc.values())))
