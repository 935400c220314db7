#!/usr/bin/env python
"""Where does the V1 decoder's error come from?  Engine taps vs oracle taps on one seeded case (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import restate_v1, synth_v1
from unidepth_amd import UniDepthV1
cfg = synth_v1.load_config_v1(); sd = synth_v1.make_synthetic_checkpoint_v1(cfg, 211)
rgb = torch.randint(0, 256, (1, 3, 240, 320), dtype=torch.uint8, generator=torch.Generator().manual_seed(5))
orc = restate_v1.OracleV1(cfg, sd); ref = orc.infer(rgb); T = orc.taps_v1
m = UniDepthV1(cfg).load_state_dict(sd).to("cuda").eval()
out, taps = m.infer_with_taps(rgb.cuda())
def rel(a, b): a, b = a.double().cpu(), b.double().cpu(); return ((a - b).norm() / b.norm()).item()
for j in range(4): print(f"features[{j}]", f"{rel(taps['features'][j], T['features'][j]):.2e}")
for k in ("rays_embedding_16", "to_latents", "aggregate_16", "prompt_camera", "latents_16", "up8", "layers_8", "up4", "layers_4", "up2", "out8", "out4", "out2"):
    a, b = taps[k], T[k]
    print(k, f"{rel(a.reshape(b.shape), b):.2e}", "  |ref| rms", f"{b.pow(2).mean().sqrt().item():.3g}")
print("depth ARel", ((out['depth'].cpu() - ref['depth']).abs() / ref['depth']).mean().item())
