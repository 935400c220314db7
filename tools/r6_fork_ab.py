#!/usr/bin/env python
"""A/B of the camera branch as a side section of the launch program (UniDepthV2._fork_camera_branch, csrc/program.cpp fork / side_end / join) against
the single-stream program: two models on the same weights in ONE process, arms interleaved; one call at a time (p50) and two requests in flight.
    python tools/r6_fork_ab.py [--batch 8] [--rounds 5]        GPU box only."""
import argparse
import os
import statistics
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import synth
from unidepth_amd import UniDepthV2
from unidepth_amd.pipeline import InferPipeline

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--calls", type=int, default=30)
args = ap.parse_args()
cfg = synth.load_config("vitl14")
sd = synth.make_synthetic_checkpoint(cfg, 125)
models = {}
for name, fork in (("single-stream", False), ("side-section", True)):
    m = UniDepthV2(cfg).load_state_dict(sd).to("cuda").eval()
    m._fork_camera_branch = fork
    models[name] = m
rgb = torch.randint(0, 256, (args.batch, 3, 518, 518), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).cuda()
outs = {k: {n: v.clone() for n, v in m.infer(rgb).items()} for k, m in models.items()}
torch.cuda.synchronize()
same = all(torch.equal(outs["single-stream"][n], outs["side-section"][n]) for n in outs["single-stream"])
print(f"bs {args.batch}: outputs of the two programs bit-identical: {same}")
res = {k: {"p50": [], "ips2": []} for k in models}
for r in range(args.rounds):
    for k, m in models.items():
        for _ in range(3):
            m.infer(rgb)
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.calls):
            t0 = time.perf_counter()
            m.infer(rgb)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        res[k]["p50"].append(statistics.median(ts))
        pipe = InferPipeline(m, depth=2)
        for _ in range(4):
            pipe.submit(rgb)
        pipe.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.calls):
            pipe.submit(rgb)
        pipe.sync(); torch.cuda.synchronize()
        res[k]["ips2"].append(args.batch * args.calls / (time.perf_counter() - t0))
for k in models:
    print(f"{k:14s}: one call p50 per round {[round(v, 3) for v in res[k]['p50']]} ms (median {statistics.median(res[k]['p50']):.3f}); "
          f"two in flight {[round(v, 1) for v in res[k]['ips2']]} images/s (median {statistics.median(res[k]['ips2']):.1f})")
