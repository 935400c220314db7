import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidepth_amd import ops
g = torch.Generator().manual_seed(0)
def run(M, N, K, hint):
    A = torch.randn(M, K, generator=g).half().cuda(); W = (torch.randn(N, K, generator=g) * K ** -0.5).half().cuda(); bias = torch.randn(N, generator=g).cuda()
    out = torch.zeros(M, N, dtype=torch.half, device="cuda")
    P = ops.Program(); P.gemm(A=A, W=W, bias=bias, M=M, N=N, K=K, lda=K, ldw=K, out=out, ldc=N, epi=ops.UD_EPI_F16, tile_hint=hint)
    for _ in range(3): P.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): P.run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
base = 11008
for N in (256, 512, 1024, 2048, 4096, 8192):
    M = base * 4096 // N
    t = run(M, N, 64, 2); t0 = run(M, N, 64, 2 | 256)
    print(f"M={M:7d} N={N:5d} (same 90 MB fp16 output, K=64, 256x256 tiles): full {t:6.1f} us, no-epilogue {t0:6.1f} us -> epilogue {t - t0:5.1f} us = {90.2e6 / (t - t0) / 1e6:.2f} TB/s")
