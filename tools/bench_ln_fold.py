#!/usr/bin/env python
"""LayerNorm fold in isolation (ViT-L/14 bs=8 shapes): per-launch time of {LayerNorm kernel, classic consumer GEMM} against
{row-statistics finalize, folded consumer GEMM}, and of the producers (fp32 accumulate) without / with the raw copy + partial sums.
Interleaved rounds, HIP events per launch.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidepth_amd import ops
D, Np, B = 1024, 1376, 8
M = B * Np
g = torch.Generator().manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).cuda()
x = rn(M, D); xn = torch.zeros(M, D, dtype=torch.half, device="cuda"); x16 = x.half()
part = torch.zeros(M, D // 64, 2, device="cuda"); stats = torch.zeros(M, 2, device="cuda")
ao = rn(M, D).half(); hid = torch.zeros(M, 4 * D, dtype=torch.half, device="cuda"); hin = rn(M, 4 * D).half()
qk = torch.zeros(M, 2 * D, dtype=torch.half, device="cuda"); vt = torch.zeros(B, 16, 64, 1408, dtype=torch.half, device="cuda")
Wq = rn(3 * D, D, sc=D ** -0.5).half(); bq = rn(3 * D); W1 = rn(4 * D, D, sc=D ** -0.5).half(); b1 = rn(4 * D)
Wp = rn(D, D, sc=D ** -0.5).half(); bp = rn(D); W2 = rn(D, 4 * D, sc=(4 * D) ** -0.5).half(); b2 = rn(D)
wsq = Wq.double().sum(1).float(); ws1 = W1.double().sum(1).float()
ops.layernorm(x=x, y=xn, rows=M, D=D, ldx=D, ldy=D, eps=1e-6, rows_per_img=M, in_rows_per_img=M, out_rows_per_img=M)
sl = x.view(M, D // 64, 64); part[..., 0] = sl.sum(-1); part[..., 1] = (sl * sl).sum(-1)
ops.row_stats_finalize(part, stats, M, D // 64, D, 1e-6)
lnc = dict(row_stats_in=stats, ln_slabs=D // 64, ln_D=D, ln_eps=1e-6)
qkv = dict(W=Wq, bias=bq, out=qk, out2=vt, M=M, N=3 * D, K=D, lda=D, ldw=D, ldc=2 * D, epi=ops.UD_EPI_QKV, vsplit=2 * D, tok_per_img=Np, kv_ld=1408, heads_v=16)
fc1 = dict(W=W1, bias=b1, out=hid, M=M, N=4 * D, K=D, lda=D, ldw=D, ldc=4 * D, epi=ops.UD_EPI_F16, act=ops.UD_ACT_GELU)
proj = dict(A=ao, W=Wp, bias=bp, out=x, M=M, N=D, K=D, lda=D, ldw=D, ldc=D, epi=ops.UD_EPI_F32, accumulate=1)
fc2 = dict(A=hin, W=W2, bias=b2, out=x, M=M, N=D, K=4 * D, lda=4 * D, ldw=4 * D, ldc=D, epi=ops.UD_EPI_F32, accumulate=1)
prod = dict(out2=x16, ldc2=D, row_stats_out=part)
tk = torch.zeros(2, M // 128 + 2, dtype=torch.int32, device="cuda")
fin = lambda k: dict(prod, row_stats_final=stats, row_stats_ticket=tk[k], ln_D=D, ln_eps=1e-6)      # + in-kernel reduction (last workgroup per row tile)
cases = {
    "layernorm": lambda: ops.layernorm(x=x, y=xn, rows=M, D=D, ldx=D, ldy=D, eps=1e-6, rows_per_img=M, in_rows_per_img=M, out_rows_per_img=M),
    "finalize": lambda: ops.row_stats_finalize(part, stats, M, D // 64, D, 1e-6),
    "qkv classic": lambda: ops.gemm(A=xn, **qkv), "qkv folded": lambda: ops.gemm(A=x16, wsum=wsq, **qkv, **lnc),
    "fc1 classic": lambda: ops.gemm(A=xn, **fc1), "fc1 folded": lambda: ops.gemm(A=x16, wsum=ws1, **fc1, **lnc),
    "fc1 classic list": lambda: ops.gemm(A=xn, tile_hint=2, **fc1), "fc1 folded list": lambda: ops.gemm(A=x16, wsum=ws1, tile_hint=2, **fc1, **lnc),
    "proj classic": lambda: ops.gemm(**proj), "proj producer": lambda: ops.gemm(**proj, **prod),
    "fc2 classic": lambda: ops.gemm(**fc2), "fc2 producer": lambda: ops.gemm(**fc2, **prod),
    "proj producer+final": lambda: ops.gemm(**proj, **fin(0)), "fc2 producer+final": lambda: ops.gemm(**fc2, **fin(1)),
}
tot = {k: 0.0 for k in cases}
R = 5
for r in range(R + 1):
    for k, fn in cases.items():
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        if r: tot[k] += e0.elapsed_time(e1) / 10 * 1e3
for k in cases: print(f"{k:20s} {tot[k] / R:7.1f} us")
