#!/usr/bin/env python
"""Is the attention kernel limited by what its operands cost in POWER?    The same launch (encoder shape B=8, H=16, N=1370,
q pre-scaled) on random fp16 data, on constant data and on zeros: the instruction stream, the memory traffic and the control flow are
identical (the deferred-maximum path is taken on tile 0 only in every case), only the toggling of the operand bits differs.
profiles/r02_mfma_attainable.txt measured 1.45x between constant and random operands on a pure MFMA stream.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidepth_amd import ops

B, H, N = 8, 16, 1370
D = H * 64; Np = 1376; kvld = 1408


def run(tag, qk, vt):
    o = torch.zeros(B * Np, D, dtype=torch.half, device="cuda")
    P = ops.Program()
    P.attention(Q=qk, K=qk.data_ptr() + D * 2, Vt=vt, O=o, B=B, H=H, Nq=N, Nk=N, ldq=2 * D, ldk=2 * D, ldo=D, kv_ld=kvld, q_rows_per_img=Np,
                k_rows_per_img=Np, scale=0.125, q_prescaled=1)
    for _ in range(5):
        P.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(4):
        e0.record()
        for _ in range(40):
            P.run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 40 * 1e3)
    fl = 4.0 * B * H * N * N * 64
    print(f"{tag:34s} {min(ts):6.1f} us  ({fl / min(ts) / 1e6:5.0f} TFLOP/s)   rounds " + " ".join(f"{t:6.1f}" for t in ts), flush=True)


g = torch.Generator().manual_seed(0)
c = 0.125 * 1.4426950408889634
rq = torch.randn(B * Np, D, generator=g) * 2.0
rk = torch.randn(B * Np, D, generator=g) * 2.0
rv = torch.randn(B, H, 64, kvld, generator=g)
cases = {
    "random q, k, v (N(0, 2))": (torch.cat([rq * c, rk], 1), rv),
    "random q, k; constant v": (torch.cat([rq * c, rk], 1), torch.full_like(rv, 0.75)),
    "constant q, k; random v": (torch.cat([torch.full_like(rq, 0.3), torch.full_like(rk, 0.6)], 1), rv),
    "constant q, k, v": (torch.cat([torch.full_like(rq, 0.3), torch.full_like(rk, 0.6)], 1), torch.full_like(rv, 0.75)),
    "zeros": (torch.zeros(B * Np, 2 * D), torch.zeros_like(rv)),
}
for rep in range(2):
    for tag, (qk, vt) in cases.items():
        run(tag, qk.half().cuda(), vt.half().cuda())

# ---- the same question for the encoder's GEMMs (warm operands, alone): fc1 (GELU, fp16 out) and fc2 (fp32 residual accumulate), random against constant operands
M = 8 * 1376


def gemm_case(tag, N, K, kind, const):
    if const:
        A = torch.full((M, K), 0.5).half().cuda(); W = torch.full((N, K), K ** -0.5).half().cuda()
    else:
        A = torch.randn(M, K, generator=g).half().cuda(); W = (torch.randn(N, K, generator=g) * K ** -0.5).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    P = ops.Program()
    if kind == "gelu":
        out = torch.zeros(M, N, dtype=torch.half, device="cuda"); kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F16, act=ops.UD_ACT_GELU)
    else:
        out = torch.zeros(M, N, device="cuda"); kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F32, accumulate=1)
    P.gemm(A=A, W=W, bias=bias, M=M, N=N, K=K, lda=K, ldw=K, **kw)
    for _ in range(5):
        P.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(4):
        e0.record()
        for _ in range(40):
            P.run()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 40 * 1e3)
    fl = 2.0 * M * N * K
    print(f"{tag:34s} {min(ts):6.1f} us  ({fl / min(ts) / 1e6:5.0f} TFLOP/s)   rounds " + " ".join(f"{t:6.1f}" for t in ts), flush=True)


for rep in range(2):
    for const in (False, True):
        gemm_case(f"fc1 4096 x 1024, {'constant' if const else 'random'} operands", 4096, 1024, "gelu", const)
        gemm_case(f"fc2 1024 x 4096, {'constant' if const else 'random'} operands", 1024, 4096, "acc", const)
