#!/usr/bin/env python
"""Attention at the encoder shape: classic (Q, scale) path vs q_prescaled path of the loaded library build.  GPU box only."""
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
res = {}
for pre in (0, 1):
    P = ops.Program()
    P.attention(Q=qk, K=qk.data_ptr() + D * 2, Vt=vt, O=o, B=B, H=H, Nq=N, Nk=N, ldq=2 * D, ldk=2 * D, ldo=D, kv_ld=kvld, q_rows_per_img=Np, k_rows_per_img=Np, scale=0.125, q_prescaled=pre)
    res[pre] = P
import statistics
ts = {0: [], 1: []}
for rnd in range(12):
    for pre, P in res.items():
        for _ in range(2): P.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): P.run()
        e1.record(); torch.cuda.synchronize()
        if rnd >= 2: ts[pre].append(e0.elapsed_time(e1) / 20 * 1e3)
for pre in (0, 1):
    us = statistics.median(ts[pre])
    print(os.environ.get("UNIDEPTH_HIP_LIB", "default").split("/")[-1], f"q_prescaled={pre}: median {us:.1f} us (min {min(ts[pre]):.1f})  {4.0 * B * H * N * N * 64 / us / 1e6:.1f} TFLOP/s")
