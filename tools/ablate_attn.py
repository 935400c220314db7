#!/usr/bin/env python
"""Ablation timings of the attention kernel at the encoder shape (instrumented build: csrc/build.sh -DUD_ABLATE with
UD_OUT=ab/libablate.so; run with UNIDEPTH_HIP_LIB=ab/libablate.so).  GPU box only.  Results of ablated variants are wrong by
construction; only their durations are read."""
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
P.attention(Q=qk, K=qk.data_ptr() + D * 2, Vt=vt, O=o, B=B, H=H, Nq=N, Nk=N, ldq=2 * D, ldk=2 * D, ldo=D, kv_ld=kvld, q_rows_per_img=Np, k_rows_per_img=Np, scale=0.125)
names = {0: "full", 1: "no softmax VALU", 2: "no PV mfma", 4: "no QK mfma", 3: "no softmax, no PV", 7: "no softmax/PV/QK (LDS + traffic + barrier only)",
         8: "no K/V traffic", 9: "no traffic, no softmax", 24: "no traffic, no barrier", 25: "no traffic/barrier/softmax (MFMA + LDS reads)", 31: "nothing but LDS reads"}
for abl, nm in names.items():
    ops.lib.ud_set_debug_flags(abl << 8)
    for _ in range(3): P.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): P.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"ABL={abl:2d} {nm:50s}: {us:7.1f} us  ({4.0 * B * H * N * N * 64 / us / 1e6:.0f} TFLOP/s-equivalent)")
ops.lib.ud_set_debug_flags(0)
