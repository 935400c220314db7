#!/usr/bin/env python
"""The four encoder GEMM shapes (ViT-L/14, bs=8) through ud_gemm_f16, interleaved rounds: for A/B-ing library builds.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unidepth_amd import ops
M = 8 * 1376
g = torch.Generator().manual_seed(0)
def mk(N, K, kind, hint=0):
    A = torch.randn(M, K, generator=g).half().cuda(); W = (torch.randn(N, K, generator=g) * K ** -0.5).half().cuda(); bias = torch.randn(N, generator=g).cuda()
    P = ops.Program()
    if kind == "gelu":
        out = torch.zeros(M, N, dtype=torch.half, device="cuda"); kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F16, act=ops.UD_ACT_GELU)
    elif kind == "acc":
        out = torch.zeros(M, N, device="cuda"); kw = dict(out=out, ldc=N, epi=ops.UD_EPI_F32, accumulate=1)
    else:
        D = N // 3
        out = torch.zeros(M, 2 * D, dtype=torch.half, device="cuda"); vt = torch.zeros(8, D // 64, 64, 1408, dtype=torch.half, device="cuda")
        kw = dict(out=out, out2=vt, ldc=2 * D, epi=ops.UD_EPI_QKV, vsplit=2 * D, tok_per_img=1376, kv_ld=1408, heads_v=D // 64)
    P.gemm(A=A, W=W, bias=bias, M=M, N=N, K=K, lda=K, ldw=K, tile_hint=hint, **kw)
    return P, 2.0 * M * N * K, (out, kw.get('out2'))
def bench(progs, hint):
  tot = {k: 0.0 for k in progs}
  R = 5
  for r in range(R + 1):
      for k, (P, fl, _) in progs.items():
          for _ in range(2): P.run()
          torch.cuda.synchronize()
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          e0.record()
          for _ in range(10): P.run()
          e1.record(); torch.cuda.synchronize()
          if r: tot[k] += e0.elapsed_time(e1) / 10 * 1e3
  print(f"hint {hint}", "  ".join(f"{k} {tot[k] / R:6.1f} us ({progs[k][1] / (tot[k] / R) / 1e6:4.0f} TF)" for k in progs), f" sum {sum(tot.values()) / R:.1f}")
  
HINTS = [int(h) for h in os.environ.get("UD_TILE_HINTS", "0").split(",")]        # 0 auto, 8 / 9 / 10: 4-wave layout auto / 256-row / 192-row
ref = {}
for hint in HINTS:
  g = torch.Generator().manual_seed(0)
  progs = {"qkv": mk(3072, 1024, "qkv", hint), "proj": mk(1024, 1024, "acc", hint), "fc1": mk(4096, 1024, "gelu", hint), "fc2": mk(1024, 4096, "acc", hint)}
  for k, (P, fl, outs) in progs.items():                      # one run from zeroed outputs: every layout must give the same bits
    P.run(); torch.cuda.synchronize()
    cur = [o.clone() for o in outs if o is not None]
    if k in ref:
        same = all(torch.equal(a, b) for a, b in zip(ref[k], cur))
        if not same: print(f"  hint {hint} {k}: DIFFERS from hint {HINTS[0]}: max abs", max(float((a.float() - b.float()).abs().max()) for a, b in zip(ref[k], cur)))
    else: ref[k] = cur
  bench(progs, hint)
