#!/usr/bin/env python
"""UniDepthV1 (ConvNeXt-L) on one MI355X at BASELINE.json configs[3]: 640x480 inputs, batch 16.  Prints ONE JSON line in the layout of
bench.py (metric / value / roofline of the dominant kernel class / cpu_baseline = the fp32 oracle on the host cores, bounded sample) and
the per-kernel-class breakdown (HIP events around every launch of the program).  GPU box only.   python tools/bench_v1.py [batch] [--vitl14] [--no-cpu] [--by-tag] [--dump]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import synth_v1
from unidepth_amd import UniDepthV1
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
ARCH = "vitl14" if "--vitl14" in sys.argv else "cnvnxtl"         # --vitl14: UniDepthV1 on the DINOv2 ViT-L/14 backbone (hubconf.py:14-17), same decoder
cfg = synth_v1.load_config_v1(*(("vitl14",) if ARCH == "vitl14" else ())); sd = synth_v1.make_synthetic_checkpoint_v1(cfg, 212 if ARCH == "vitl14" else 211)
m = UniDepthV1(cfg).load_state_dict(sd).to("cuda").eval()
rgb = torch.randint(0, 256, (B, 3, 480, 640), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).cuda()
for _ in range(3): m.infer(rgb)
torch.cuda.synchronize()
t0 = time.perf_counter(); K = 10
for _ in range(K): m.infer(rgb)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
plan = next(reversed(m._plans.values())); P = plan.prog; n = len(P)
tot = {}
bytag = {}
for rep in range(2):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record()
    for i in range(n):
        P.run(i, i + 1); evs[i + 1].record()
    torch.cuda.synchronize()
    if rep:
        for i in range(n):
            cls, tag, fl, nb = P.meta[i]
            key = cls if cls.startswith("gemm") or cls.startswith("conv") else tag if cls.startswith("v1.") else cls
            d = tot.setdefault(key, [0.0, 0.0, 0]); d[0] += evs[i].elapsed_time(evs[i + 1]); d[1] += fl; d[2] += 1
            d = bytag.setdefault(("enc " if i < plan.dec_first else "dec ") + tag, [0.0, 0.0, 0]); d[0] += evs[i].elapsed_time(evs[i + 1]); d[1] += fl; d[2] += 1
enc_ms = sum(evs[i].elapsed_time(evs[i + 1]) for i in range(plan.dec_first))
dom, dv = max(((k, v) for k, v in tot.items() if v[1] > 0), key=lambda kv: kv[1][0])
line = {"metric": "images/sec, UniDepthV1 ConvNeXt-L 640x480 (BASELINE configs[3])" if ARCH == "cnvnxtl" else "images/sec, UniDepthV1 ViT-L/14 640x480", "value": round(B / dt, 2), "unit": "images/s", "n_gpus": 1,
        "ms_per_step": round(dt * 1e3, 3), "higher_is_better": True, "dtype": "f16", "data": "synthetic uint8 RGB (seeded) resident in HBM; seeded random-init weights",
        "config": {"workload": f"UniDepthV1 {'ConvNeXt-L' if ARCH == 'cnvnxtl' else 'ViT-L/14'} infer(), 640x480, bs={B}, one call at a time"},
        "launches": n, "encoder_ms": round(enc_ms, 3), "decoder_ms": round(sum(v[0] for v in tot.values()) - enc_ms, 3),
        "roofline": {"bound": "mfma", "kernel": dom + " (v_mfma_f32_16x16x32_f16)", "achieved": round(dv[1] / (dv[0] * 1e-3) / 1e12, 2), "peak": 2500.0,
                     "unit": "TFLOP/s", "frac": round(dv[1] / (dv[0] * 1e-3) / 1e12 / 2500.0, 4), "avg_launch_us": round(dv[0] * 1e3 / dv[2], 2),
                     "flop_per_launch": round(dv[1] / dv[2], 1), "traffic": None}}
if "--no-cpu" not in sys.argv:
    from oracle import restate_v1                      # CPU baseline leg only: the checker, timed on the host cores
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    orc = restate_v1.OracleV1(cfg, sd)
    one = rgb[:1].cpu()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); orc.infer(one); ts.append(time.perf_counter() - t0)
    p50 = sorted(ts)[1]
    line["cpu_baseline"] = {"value": round(1.0 / p50, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port", "p50_s": round(p50, 3),
                            "sample": "oracle/restate_v1.py fp32, ONE 640x480 image of the bench batch, 3 passes, p50"}
print(json.dumps(line))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"  {k:58s} {v[0]:8.3f} ms  x{v[2]:4d}  {v[1] / (v[0] * 1e-3) / 1e12 if v[1] else 0:7.1f} TF")
if "--dump" in sys.argv:                        # every launch of the decoder in program order: index, class, tag, us
    for i in range(plan.dec_first, n):
        cls, tag, fl, nb = P.meta[i]
        print(f"  #{i:4d} {cls[:44]:44s} {tag:28s} {evs[i].elapsed_time(evs[i + 1]) * 1e3:8.1f} us")
if "--by-tag" in sys.argv:                      # the same launches by program tag (which layer), largest first
    for k, v in sorted(bytag.items(), key=lambda kv: -kv[1][0])[:70]:
        print(f"  {k:58s} {v[0]:8.3f} ms  x{v[2]:4d}  {v[1] / (v[0] * 1e-3) / 1e12 if v[1] else 0:7.1f} TF")
