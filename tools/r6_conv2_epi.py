#!/usr/bin/env python
"""Round 6: where does the residual epilogue of the decoder's second RCU convolution go?  The stage-2 launches (M = 175232, N = 256, K = 2304) run the same MFMA
work as conv1 (fp16 + LeakyReLU epilogue) but take 225-264 us against 191.  Variants of the SAME launch, interleaved in one process:
  full      out(fp32) += ..., fp16 copy               (what runs: 179 MB read + 179 MB + 90 MB written)
  copyonly  accumulate = 2: fp32 read, only the fp16 copy written
  nopre     accumulate = 0: no old values read, fp32 + fp16 written
  f32only   accumulate = 0, no fp16 copy
  f16       the fp16 epilogue of conv1 on the same operands
GPU box only.   python tools/r6_conv2_epi.py [--match dh.ups.2.0.conv2]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import synth
from unidepth_amd import UniDepthV2, ops

ap = argparse.ArgumentParser()
ap.add_argument("--match", default="dh.ups.2.0.conv2,dh.ups.1.0.conv2")
ap.add_argument("--rounds", type=int, default=4)
args = ap.parse_args()
captured = []
orig = ops.Program.gemm
def spy(self, **kw):
    captured.append(dict(kw)); return orig(self, **kw)
ops.Program.gemm = spy
cfg = synth.load_config("vitl14")
model = UniDepthV2(cfg).load_state_dict(synth.make_synthetic_checkpoint(cfg, 125)).to("cuda").eval()
rgb = torch.randint(0, 256, (8, 3, 518, 518), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).cuda()
model.infer(rgb); torch.cuda.synchronize()
ops.Program.gemm = orig
names = {v.data_ptr(): k for k, v in model._w.items() if isinstance(v, torch.Tensor) and v.is_cuda}
for kw in captured:
    name = kw.get("tag") or names.get(kw["W"].data_ptr() if isinstance(kw["W"], torch.Tensor) else kw["W"], "?")
    if not any(m in name for m in args.match.split(",")):
        continue
    kw = {k: v for k, v in kw.items() if k not in ("tag", "flops", "splitk_ws", "splitk_cnt", "splitk_ws_bytes")}
    M, N = kw["M"], kw["N"]
    o16 = torch.zeros(M, N, dtype=torch.half, device="cuda")
    var = {
        "full": dict(kw, accumulate=1),
        "copyonly": dict(kw, accumulate=2),
        "nopre": dict(kw, accumulate=0),
        "f32only": {k: v for k, v in dict(kw, accumulate=0).items() if k not in ("out2", "ldc2", "act2")},
        "f16": {k: v for k, v in dict(kw, out=o16, epi=ops.UD_EPI_F16, act=ops.UD_ACT_LRELU, accumulate=0).items() if k not in ("out2", "ldc2", "act2")},
    }
    progs = {}
    for k, d in var.items():
        P = ops.Program(); P.gemm(**d); progs[k] = P
    tot = {k: 0.0 for k in progs}
    for r in range(args.rounds + 1):
        for k, P in progs.items():
            P.run(); P.run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): P.run()
            e1.record(); torch.cuda.synchronize()
            if r: tot[k] += e0.elapsed_time(e1) / 10 * 1e3 / args.rounds
    print(f"{name} M {M} N {N} K {kw['K']}: " + "  ".join(f"{k} {tot[k]:6.1f} us [{progs[k].meta[0][0][:40]}]" for k in progs), flush=True)
