#!/usr/bin/env python
"""A/B of tile schedules on the REAL launches of a plan (ViT-L/14 518x518 bs=8 by default): the GEMM descriptors are captured while the plan is
recorded, every selected launch is re-recorded as a one-op program per `tile_hint` and the arms are timed interleaved in one process; results of
the arms are compared (max abs / rel difference against the first arm; split-K changes the summation order, so not bit-equal).
    python tools/r6_dec_ab.py --match dh.ups.0 --hints 1,3,10        GPU box only."""
import argparse
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import synth
from unidepth_amd import UniDepthV2, ops

ap = argparse.ArgumentParser()
ap.add_argument("--match", default="dh.ups.0", help="comma-separated substrings of the weight name / tag")
ap.add_argument("--hints", default="1,3,10")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--rounds", type=int, default=4)
args = ap.parse_args()

captured = []
orig = ops.Program.gemm


def spy(self, **kw):
    captured.append(dict(kw))
    return orig(self, **kw)


ops.Program.gemm = spy
cfg = synth.load_config("vitl14")
model = UniDepthV2(cfg).load_state_dict(synth.make_synthetic_checkpoint(cfg, 125)).to("cuda").eval()
rgb = torch.randint(0, 256, (args.batch, 3, 518, 518), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).cuda()
model.infer(rgb)
torch.cuda.synchronize()
ops.Program.gemm = orig
names = {v.data_ptr(): k for k, v in model._w.items() if isinstance(v, torch.Tensor) and v.is_cuda}
hints = [int(h) for h in args.hints.split(",")]
sel = []
for kw in captured:
    name = kw.get("tag") or names.get(kw["W"].data_ptr() if isinstance(kw["W"], torch.Tensor) else kw["W"], "?")
    if any(m in name for m in args.match.split(",")):
        sel.append((name, kw))
print(f"{len(captured)} GEMM launches captured, {len(sel)} selected")
for name, kw in sel:
    kw = {k: v for k, v in kw.items() if k not in ("tag", "flops", "splitk_ws", "splitk_cnt", "splitk_ws_bytes")}
    outs = [kw[k] for k in ("out", "out2") if isinstance(kw.get(k), torch.Tensor)]
    snap = [o.clone() for o in outs]
    progs, res = {}, {}
    for h in hints:
        P = ops.Program()
        try:
            P.gemm(**dict(kw, tile_hint=h))
        except Exception as e:
            print(f"  {name} hint {h}: {e}")
            continue
        progs[h] = P
        for o, s in zip(outs, snap):
            o.copy_(s)
        P.run()
        torch.cuda.synchronize()
        res[h] = [o.float().clone() for o in outs]
    base = next(iter(res))
    diffs = {h: max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(res[h], res[base])) for h in res}
    tot = {h: 0.0 for h in progs}
    for r in range(args.rounds + 1):
        for h, P in progs.items():
            P.run(); P.run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                P.run()
            e1.record(); torch.cuda.synchronize()
            if r:
                tot[h] += e0.elapsed_time(e1) / 10 * 1e3 / args.rounds
    fl = 2.0 * kw["M"] * kw["N"] * kw["K"] * max(1, kw.get("groups", 0))
    print(f"{name:28s} M {kw['M']} N {kw['N']} K {kw['K']} g {kw.get('groups', 0)} epi {kw.get('epi', 0)}: " +
          " | ".join(f"hint {h}: {tot[h]:6.1f} us {fl / tot[h] / 1e6:5.0f} TF [{progs[h].meta[0][0][:44]}] d {diffs[h]:.1e}" for h in progs), flush=True)
    for o, s in zip(outs, snap):
        o.copy_(s)
