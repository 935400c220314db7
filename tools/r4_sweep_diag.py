#!/usr/bin/env python
"""Why does a checkpoint seed miss the 1e-3 bar?  Per-tap errors of UniDepthV2 ViT-L against the oracle for one (seed, size), with the
engine's camera replaced by the oracle's K as a second run (separates camera-head error from the depth stack's own).  GPU box only.
usage: r4_sweep_diag.py seed H W"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import restate, synth
from unidepth_amd import UniDepthV2
seed, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = synth.load_config("vitl14")
sd = synth.make_synthetic_checkpoint(cfg, seed)
rgb = torch.randint(0, 256, (1, 3, H, W), dtype=torch.uint8, generator=torch.Generator().manual_seed(seed + H))
orc = restate.OracleV2(cfg, sd); orc.keep_taps = True
ref = orc.infer(rgb); rt = orc.taps
model = UniDepthV2(cfg).load_state_dict(sd).to("cuda").eval()
out, taps = model.infer_with_taps(rgb.cuda())
torch.cuda.synchronize()
rel = lambda a, b: ((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
arel = lambda o, r: ((o["depth"].float().cpu() - r["depth"]).abs() / r["depth"]).mean().item()
print(f"seed {seed} {H}x{W}: depth ARel {arel(out, ref):.2e}  K {((out['intrinsics'].cpu() - ref['intrinsics']).abs() / ref['intrinsics'].abs().clamp_min(1.0)).max().item():.2e}")
print("intrinsics4 engine", taps["intrinsics4"].cpu().tolist(), "oracle", rt["intrinsics4"].tolist())
for name in taps:
    if name in rt and torch.is_tensor(rt[name]) and rt[name].numel() == taps[name].numel():
        print(f"  {name:28s} rel-L2 {rel(taps[name].float().reshape(rt[name].shape), rt[name]):.2e}   |ref| max {rt[name].abs().max().item():.3g}")
for name in ("logdepth", "logconf"):
    d = (taps[name].cpu() - rt[name]).abs()
    print(f"  {name}: |d| mean {d.mean().item():.2e} max {d.max().item():.2e}; signed mean {(taps[name].cpu() - rt[name]).mean().item():+.2e}")
# the same image with the ORACLE's predicted camera given as GT: the depth stack alone
K = ref["intrinsics"].clone()
out2 = model.infer(rgb.cuda(), K)
ref2 = orc.infer(rgb, K.clone())
torch.cuda.synchronize()
print(f"with the oracle's K as GT camera: depth ARel {arel(out2, ref2):.2e}")
# ---- camera path in detail: the engine's final-normed class tokens against the oracle's, and the ORACLE's camera head fed with them
dbg = model.debug_taps()
_ = model.infer(rgb.cuda())            # (debug_taps reads the last plain call's plan)
torch.cuda.synchronize()
dbg = model.debug_taps()
img_n = None
feats_o, cls_o = None, None
import types
orc2 = restate.OracleV2(cfg, sd)
_enc = orc2.encode
store = {}
def enc_hook(image):
    f, c = _enc(image)
    store["cls"] = c
    return f, c
orc2.encode = enc_hook
_ = orc2.infer(rgb)
cls_e = [c.float().cpu() for c in dbg["tokens"]]
for j, (ce, co) in enumerate(zip(cls_e, store["cls"])):
    print(f"  final-normed class token, level {j}: rel-L2 {rel(ce, co):.2e}  max|d| {(ce - co).abs().max().item():.2e}  |ref| max {co.abs().max().item():.3g}")
def camera(o, cls):
    ct = torch.cat([o._lin(x, f"pixel_decoder.camera_token_adapter.input_adapters.{j}") for j, x in enumerate(cls)], dim=1)
    return o._camera_head(ct, model._plans[next(reversed(model._plans))].Hn, model._plans[next(reversed(model._plans))].Wn)
k_or = camera(orc2, store["cls"])
k_mix = camera(orc2, cls_e)
k_en = taps["intrinsics4"].cpu()
print("K oracle                         ", k_or[0].tolist())
print("K oracle head on engine cls      ", k_mix[0].tolist(), " max-rel vs oracle", ((k_mix - k_or).abs() / k_or.abs()).max().item())
print("K engine                         ", k_en[0].tolist(), " max-rel vs oracle", ((k_en - k_or).abs() / k_or.abs()).max().item(),
      " vs oracle-head-on-engine-cls", ((k_en - k_mix).abs() / k_mix.abs()).max().item())
if len(sys.argv) > 4:                  # the headline batch: 8 images, image 0 = this one (LayerNorm fold on)
    rgb8 = torch.cat([rgb] + [torch.randint(0, 256, (1, 3, H, W), dtype=torch.uint8, generator=torch.Generator().manual_seed(seed + 7 * i)) for i in range(1, 8)])
    o8 = model.infer(rgb8.cuda())
    torch.cuda.synchronize()
    d0 = ((o8["depth"][:1].float().cpu() - ref["depth"]).abs() / ref["depth"]).mean().item()
    k0 = ((o8["intrinsics"][:1].cpu() - ref["intrinsics"]).abs() / ref["intrinsics"].abs().clamp_min(1.0)).max().item()
    print(f"same image as image 0 of a batch of 8: depth ARel {d0:.2e}  K {k0:.2e}")
