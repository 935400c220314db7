#!/usr/bin/env python
"""BASELINE.json configs[4] on one GPU: 16 x 644x966 + 16 x 518x518 images through dist.infer_mixed (ViT-L/14), with and without
overlapping micro-batches.  GPU box only; not the bench.py metric."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
warnings.simplefilter("ignore")
from oracle import synth
from unidepth_amd import UniDepthV2
from unidepth_amd.dist import infer_mixed
cfg = synth.load_config("vitl14")
model = UniDepthV2(cfg).load_state_dict(synth.make_synthetic_checkpoint(cfg, 125)).to("cuda").eval()
g = torch.Generator().manual_seed(1)
imgs = [torch.randint(0, 256, (3, 644, 966), dtype=torch.uint8, generator=g).cuda() for _ in range(16)] + \
       [torch.randint(0, 256, (3, 518, 518), dtype=torch.uint8, generator=g).cuda() for _ in range(16)]
for inflight in (1, 2, 1, 2):
    for _ in range(2): infer_mixed(model, imgs, inflight=inflight)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): infer_mixed(model, imgs, inflight=inflight)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"mixed 16x(644x966) + 16x(518x518), inflight={inflight}: {dt * 1e3:.1f} ms per 32 images = {32 / dt:.1f} images/s")
