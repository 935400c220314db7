import os, sys
sys.path.insert(0, "/root/repo")
import torch
from unidepth_amd import ops
M, N, K = 16384, 4096, 4096
g = torch.Generator().manual_seed(0)
A = (torch.rand(M, K, generator=g) - 0.5).half().cuda(); W = ((torch.rand(N, K, generator=g) - 0.5) * 0.1).half().cuda(); bias = torch.zeros(N).cuda()
for hint in (2, 3):
    out = torch.zeros(M, N, dtype=torch.half, device="cuda")
    P = ops.Program(); P.gemm(A=A, W=W, bias=bias, out=out, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, epi=ops.UD_EPI_F16, tile_hint=hint)
    for _ in range(2): P.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): P.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"product 8-wave 16x16x32 kernel, tile_hint {hint}: {us:.1f} us, {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s")
