#!/usr/bin/env python
"""Per-workgroup timeline of the large-tile GEMM on the encoder shapes (instrumented build: csrc/build.sh -DUD_TRACE [-DUD_TRACE_DRAIN]
with UD_OUT=ab/libtrace.so; run with UNIDEPTH_HIP_LIB=ab/libtrace.so).  GPU box only.
Stamps (100 MHz wall clock) per (workgroup, tile): 0 tile start, 1 first K-tile landed, 2 K loop done, 3 epilogue issued,
4 stores drained (UD_TRACE_DRAIN builds), 5 end-of-tile barrier."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unidepth_amd import ops

M = 8 * 1376
g = torch.Generator().manual_seed(0)
trace = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
ops.lib.ud_trace_set.argtypes = [C.c_void_p]
assert ops.lib.ud_trace_set(trace.data_ptr()) == 0


def run(name, N, K, kind, hint=0):
    A = torch.randn(M, K, generator=g).half().cuda()
    W = (torch.randn(N, K, generator=g) * K ** -0.5).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    kw = {}
    if kind == "gelu":
        out = torch.zeros(M, N, dtype=torch.half, device="cuda"); kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F16, act=ops.UD_ACT_GELU)
    elif kind == "acc":
        out = torch.zeros(M, N, device="cuda"); kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F32, accumulate=1)
    elif kind == "qkv":
        D = N // 3
        out = torch.zeros(M, 2 * D, dtype=torch.half, device="cuda"); vt = torch.zeros(8, D // 64, 64, 1408, dtype=torch.half, device="cuda")
        kw = dict(out=out, out2=vt, ldc=2 * D, epi=ops.UD_EPI_QKV, vsplit=2 * D, tok_per_img=1376, kv_ld=1408, heads_v=D // 64)
    P = ops.Program()
    P.gemm(A=A, W=W, bias=bias, M=M, N=N, K=K, lda=K, ldw=K, tile_hint=hint, **kw)
    for _ in range(3):
        P.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        P.run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    trace.zero_()
    P.run()
    torch.cuda.synchronize()
    t = trace.cpu().view(256, 8, 8).double() * 0.01      # us
    t0 = t[:, 0, 0][t[:, 0, 0] > 0].min()
    print(f"== {name}: M={M} N={N} K={K} {kind}: {us:.1f} us/launch, {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s ({P.meta[0][0]})")
    for ti in range(8):
        v = t[:, ti, 0] > 0
        if not v.any():
            break
        x = t[v, ti] - t0
        seg = lambda a, b: (x[:, b] - x[:, a])
        line = (f"  tile {ti}: n={int(v.sum()):3d} start {x[:,0].mean():6.1f} (max {x[:,0].max():6.1f}) | prologue {seg(0,1).mean():5.2f} (max {seg(0,1).max():5.2f})"
                f" | kloop {seg(1,2).mean():6.2f} (min {seg(1,2).min():6.2f} max {seg(1,2).max():6.2f}) | epilogue {seg(2,3).mean():5.2f} (max {seg(2,3).max():5.2f})")
        if (x[:, 4] > 0).any():
            line += f" | drain {seg(3,4).mean():5.2f} (max {seg(3,4).max():5.2f})"
        line += f" | end {x[:,5].mean():6.1f} (max {x[:,5].max():6.1f})"
        print(line)


run("qkv", 3072, 1024, "qkv")
run("proj", 1024, 1024, "acc")
run("fc1", 4096, 1024, "gelu")
run("fc2", 1024, 4096, "acc")
run("fc1@192", 4096, 1024, "gelu", hint=3)
run("proj@256", 1024, 1024, "acc", hint=2)
