#!/usr/bin/env python
"""Where does a key tile's time go inside the attention kernel?  Needs the stamped build (tools/r4_attn_variants.sh trace="-DUD_ATTN_TRACE=1",
UNIDEPTH_HIP_LIB=ab/libattn_trace.so): per-wave shader-clock sums between seven points of the tile loop, averaged per wave-tile.
Encoder shape (B=8, H=16, N=1370, q pre-scaled).  GPU box only."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidepth_amd import ops
B, H, N = 8, 16, 1370
D = H * 64; Np = 1376; kvld = 1408
g = torch.Generator().manual_seed(0)
qk = (torch.randn(B * Np, 2 * D, generator=g) * 2.0)
qk[:, :D] *= 0.125 * 1.4426950408889634
qk = qk.half().cuda()
vt = torch.randn(B, H, 64, kvld, generator=g).half().cuda()
o = torch.zeros(B * Np, D, dtype=torch.half, device="cuda")
tr = torch.zeros(8, dtype=torch.int64, device="cuda")
ops.lib.ud_attn_trace_set.argtypes = [C.c_void_p]
assert ops.lib.ud_attn_trace_set(tr.data_ptr()) == 0
P = ops.Program()
P.attention(Q=qk, K=qk.data_ptr() + D * 2, Vt=vt, O=o, B=B, H=H, Nq=N, Nk=N, ldq=2 * D, ldk=2 * D, ldo=D, kv_ld=kvld, q_rows_per_img=Np, k_rows_per_img=Np, scale=0.125, q_prescaled=1)
for _ in range(3): P.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tr.zero_()
e0.record()
for _ in range(10): P.run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
t = tr.cpu().double()
n = t[6].item()
names = ["DMA issue (+ ring bookkeeping)", "K reads + QK^T MFMAs (until scores readable)", "softmax VALU (max, exchange, exp, sum, pack)", "V^T reads + PV MFMAs issued",
         "vmcnt(0): own DMA pieces of next tile", "s_barrier"]
tot = t[:6].sum().item() / n
print(f"stamped kernel: {us:.1f} us per launch; {n / 10:.0f} wave-tiles per launch; {tot:.0f} s_memtime ticks per wave-tile (sum of segments)")
for i, nm in enumerate(names):
    v = t[i].item() / n
    print(f"  {nm:50s} {v:8.1f} ticks  {100 * v / tot:5.1f} %")
print(f"  kernel wall per wave-tile per SIMD slot: {us * 1e3 / (n / 10 / 1024):.0f} ns x 4 waves per SIMD = {4 * us * 1e3 / (n / 10 / 1024):.0f} ns per tile of one wave")
