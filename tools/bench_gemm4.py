import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidepth_amd import ops
M = 8 * 1376
g = torch.Generator().manual_seed(0)
def run(N, K, kind, hint):
    A = torch.randn(M, K, generator=g).half().cuda(); W = (torch.randn(N, K, generator=g) * K ** -0.5).half().cuda(); bias = torch.randn(N, generator=g).cuda()
    if kind == "f16": out = torch.zeros(M, N, dtype=torch.half, device="cuda"); kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F16)
    else: out = torch.zeros(M, N, device="cuda"); kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F32, accumulate=1)
    P = ops.Program(); P.gemm(A=A, W=W, bias=bias, M=M, N=N, K=K, lda=K, ldw=K, tile_hint=hint, **kw)
    for _ in range(3): P.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): P.run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
for N, kind in ((4096, "f16"), (1024, "acc")):
    for K in (1024, 4096):
        for base in (2, 3):
            full = run(N, K, kind, base); noepi = run(N, K, kind, base | 256); noloop = run(N, K, kind, base | 512); neither = run(N, K, kind, base | 768); nostage = run(N, K, kind, base | 1024)
            print(f"N={N} {kind} K={K} tiles={'256' if base==2 else '192'}: full {full:6.1f}  no-epilogue {noepi:6.1f}  1-ktile+epilogue {noloop:6.1f}  1-ktile,no-epilogue {neither:6.1f}  direct-stores {nostage:6.1f}")
