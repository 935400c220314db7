#!/usr/bin/env python
"""Experiment: one bs=8 infer() vs two concurrent bs=4 infer() calls on two HIP streams (do the streams fill each other's
tile-quantisation tails?).  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, warnings
warnings.simplefilter("ignore")
from oracle import synth
from unidepth_amd import UniDepthV2
cfg = synth.load_config("vitl14")
sd = synth.make_synthetic_checkpoint(cfg, 125)
model = UniDepthV2(cfg).load_state_dict(sd).to("cuda").eval()
model2 = UniDepthV2(cfg).load_state_dict(sd).to("cuda").eval()      # own plan / buffers for the second stream
g = torch.Generator().manual_seed(1)
rgb = torch.randint(0, 256, (8, 3, 518, 518), dtype=torch.uint8, generator=g).cuda()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def one():
    return model.infer(rgb)

def two():
    with torch.cuda.stream(s1):
        a = model.infer(rgb[:4])
    with torch.cuda.stream(s2):
        b = model2.infer(rgb[4:])
    return a, b

def four_seq():
    a = model.infer(rgb[:4]); b = model.infer(rgb[4:]); return a, b

for name, fn in (("one bs=8", one), ("two streams bs=4+4", two), ("sequential bs=4, bs=4", four_seq), ("one bs=8", one), ("two streams bs=4+4", two)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10 * 1e3
    print(f"{name:24s}: {dt:6.2f} ms per 8 images  ({8 / dt * 1e3:.0f} img/s)")

rgb2 = torch.randint(0, 256, (8, 3, 518, 518), dtype=torch.uint8, generator=g).cuda()
def two8():
    with torch.cuda.stream(s1):
        a = model.infer(rgb)
    with torch.cuda.stream(s2):
        b = model2.infer(rgb2)
    return a, b
for name, fn in (("two streams bs=8+8", two8), ("one bs=8 x2", lambda: (one(), one()))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10 * 1e3
    print(f"{name:24s}: {dt:6.2f} ms per 16 images  ({16 / dt * 1e3:.0f} img/s)")
