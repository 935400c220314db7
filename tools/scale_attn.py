#!/usr/bin/env python
"""Attention kernel time vs number of keys (fixed 1370 queries): separates the per-workgroup fixed cost from the per-KV-tile cost.  GPU box only."""
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
res = []
for Nk in (64, 128, 256, 512, 704, 1024, 1370):
    P = ops.Program()
    P.attention(Q=qk, K=qk.data_ptr() + D * 2, Vt=vt, O=o, B=B, H=H, Nq=N, Nk=Nk, ldq=2 * D, ldk=2 * D, ldo=D, kv_ld=kvld, q_rows_per_img=Np, k_rows_per_img=Np, scale=0.125)
    for _ in range(3): P.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): P.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    res.append((Nk, us))
    print(f"Nk={Nk:5d} ({(Nk + 63) // 64:2d} tiles): {us:7.1f} us")
(n0, t0), (n1, t1) = res[2], res[-1]
per_tile = (t1 - t0) / ((n1 + 63) // 64 - (n0 + 63) // 64)
print(f"per KV tile: {per_tile:.2f} us; fixed: {t1 - per_tile * ((n1 + 63) // 64):.1f} us")
