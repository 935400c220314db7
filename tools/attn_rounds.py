#!/usr/bin/env python
"""Attention: how much the partial second round of workgroups costs.  Encoder shape (H = 16, N = 1370: 11 query tiles per (image, head)), number
of images varied so that the grid is ~1.0 / 1.375 / 2.0 / 2.75 rounds of the 1024 workgroup slots (4 per CU).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidepth_amd import ops
H, N = 16, 1370
D = H * 64; Np = 1376; kvld = 1408
g = torch.Generator().manual_seed(0)
for pairs in (93, 128, 186, 256):
    B = 1; Hh = pairs                       # B * H pairs: use B = 1 with `pairs` heads of 64 (same work map: pair = image * H + head)
    qk = torch.randn(Np, 2 * Hh * 64, generator=g).half().cuda()
    vt = torch.randn(1, Hh, 64, kvld, generator=g).half().cuda()
    o = torch.zeros(Np, Hh * 64, dtype=torch.half, device="cuda")
    P = ops.Program()
    P.attention(Q=qk, K=qk.data_ptr() + Hh * 64 * 2, Vt=vt, O=o, B=1, H=Hh, Nq=N, Nk=N, ldq=2 * Hh * 64, ldk=2 * Hh * 64, ldo=Hh * 64, kv_ld=kvld,
                q_rows_per_img=Np, k_rows_per_img=Np, scale=0.125, q_prescaled=1)
    for _ in range(3): P.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): P.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    wgs = pairs * 11
    print(f"pairs {pairs:4d}  workgroups {wgs:5d} = {wgs / 1024:.3f} rounds  {us:7.1f} us  {us / (wgs / 1024):6.1f} us per round-equivalent  {4.0 * pairs * N * N * 64 / us / 1e6:6.1f} TFLOP/s")
