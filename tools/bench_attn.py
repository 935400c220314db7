#!/usr/bin/env python
"""Attention micro-benchmark at the encoder shape (B=8, H=16, N=1370).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidepth_amd import ops
B, H, N = 8, 16, 1370
D = H * 64; Np = 1376; kvld = 1408
g = torch.Generator().manual_seed(0)
qk = torch.randn(B * Np, 2 * D, generator=g).half().cuda()
vt = torch.randn(B, H, 64, kvld, generator=g).half().cuda()
o = torch.zeros(B * Np, D, dtype=torch.half, device="cuda")
P = ops.Program()
P.attention(Q=qk, K=qk.data_ptr() + D * 2, Vt=vt, O=o, B=B, H=H, Nq=N, Nk=N, ldq=2 * D, ldk=2 * D, ldo=D, kv_ld=kvld, q_rows_per_img=Np, k_rows_per_img=Np, scale=0.125,
            q_prescaled=int(os.environ.get("UD_ATTN_PRE", "1")))     # 1 = the engine's default: scale * log2(e) already folded into Q
for _ in range(3): P.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): P.run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"attention B={B} H={H} N={N}: {us:.1f} us  {4.0 * B * H * N * N * 64 / us / 1e6:.1f} TFLOP/s")
