#!/usr/bin/env python
"""bench.py -- throughput of the UniDepthV2 infer() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched by torch.distributed.run, one rank/GPU)

Workload (BASELINE.json metric / configs[1]): UniDepthV2 ViT-L/14, 518x518, batch 8 per GPU, synthetic uint8 RGB already
resident in HBM, seeded random-init ("sensitised") weights of the exact architecture; a step = one full infer()
(pre-process -> 24-block encoder -> decoder -> all 7 outputs on device).  With N GPUs every rank runs its own batch
(weak scaling, images are independent) and the per-step outputs `depth`, `confidence`, `intrinsics` are all-gathered
over RCCL; value = N * 8 images * K / max-over-ranks wall time.

One JSON line on rank 0 with the driver contract fields plus
  roofline     -- the dominant kernel class of the step (an MFMA GEMM instantiation, named as rocprofv3 prints it): algorithmic
                  FLOP per launch / average launch duration measured live with HIP events on the launch stream, against the
                  2.5 PFLOP/s dense fp16 peak; `traffic` = HBM bytes per launch from the committed PMC passes (profiles/);
  cpu_baseline -- the CPU oracle (fp32 restatement of the reference, oracle/restate.py) timed on this box's host cores on
                  a bounded sample (one bs=8 pass of the same workload, ~15 s), rank 0 / N=1 only.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0      # dense fp16/bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def flops_per_image(D=1024, depth=24, N=1370, C=512, hw=1369):
    """SURVEY.md 8(d) algorithmic work model (2 FLOP/MAC; softmax/LN/GELU not counted) for ViT-L/14 @ 518^2."""
    enc_gemm = depth * 2 * N * D * 12 * D
    enc_attn = depth * 4 * N * N * D
    return dict(enc_gemm=enc_gemm, enc_attn=enc_attn, enc=enc_gemm + enc_attn, total=1366.8e9)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--arch", default="vitl14")
    ap.add_argument("--size", type=int, nargs=2, default=[518, 518])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--gather", default="depth,confidence,intrinsics")
    ap.add_argument("--dump-ops", default="", help="write per-launch timings (tsv) to this file")
    ap.add_argument("--inflight", type=int, default=2, help="infer() calls in flight per GPU during the timed steps (1 = one call at a time)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("UD_BENCH_BACKEND", "nccl")      # "gloo": functional check of the N > 1 path on a box with one GPU
    if os.environ.get("UD_BENCH_SHARE_GPU"):                 # (all ranks on cuda:0; never a measurement)
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))      # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    from oracle import synth                      # seeded synthetic checkpoint generator (test infra; weights only)
    from unidepth_amd import UniDepthV2
    import warnings
    warnings.simplefilter("ignore")

    cfg = synth.load_config(args.arch)
    sd = synth.make_synthetic_checkpoint(cfg, 125)
    model = UniDepthV2(cfg).load_state_dict(sd).to(dev).eval()
    B, (H, W) = args.batch, args.size
    g = torch.Generator().manual_seed(1 + rank)
    rgb = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=g).to(dev)
    gather_keys = [k for k in args.gather.split(",") if k]

    from unidepth_amd.pipeline import InferPipeline
    pipe = InferPipeline(model, depth=max(1, args.inflight))

    def gather(out):
        # one exchange step: requested outputs packed per image, ONE RCCL all-gather (xGMI is point-to-point: few, larger messages)
        packed = torch.cat([out[k].reshape(B, -1) for k in gather_keys], dim=1)
        gathered = torch.empty((world * B, packed.shape[1]), dtype=packed.dtype, device=dev)
        if backend == "nccl":
            dist.all_gather_into_tensor(gathered, packed)
        else:
            dist.all_gather(list(gathered.chunk(world)), packed)
        return gathered

    def step():
        """One full infer() of the batch; up to --inflight steps overlap on separate HIP streams (independent batches of a
        stream of requests).  The all-gather of a step is issued on that step's stream, right behind its outputs."""
        return pipe.submit(rgb, post=gather if world > 1 else None)

    def step_single():
        out = model.infer(rgb)
        if world > 1:
            gather(out)
        return out

    for _ in range(args.warmup):
        step()
    pipe.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    pipe.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # latency of ONE call with nothing else in flight (outside the timed region)
    lat = []
    for _ in range(min(args.steps, 20)):
        ts = time.perf_counter()
        step_single()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - ts) * 1e3)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    result = {
        "metric": "images/sec (whole node) + p50 latency, ViT-L/14 518x518 bs=8",
        "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "p50_latency_ms": round(statistics.median(lat), 4),
        "inflight": max(1, args.inflight),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
        "data": "synthetic uint8 RGB (seeded) resident in HBM; seeded random-init weights of the named architecture",
        "config": {"workload": f"UniDepthV2 {args.arch} infer(), {H}x{W}, bs={B} per GPU, all 7 outputs on device; "
                               f"{max(1, args.inflight)} independent infer() calls in flight per GPU (HIP streams), p50_latency_ms = one call alone",
                   "global_batch": world * B, "parallelism": f"dp{world}" + (f" + RCCL all-gather({','.join(gather_keys)})" if world > 1 else "")},
    }

    if rank == 0:
        fl = flops_per_image()
        result["model_tflops_per_s"] = round(value * fl["total"] / 1e12 / world, 2)
        if not args.no_kernel_timing:
            result.update(kernel_timing(model, fl, B, args.dump_ops))
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, sd, H, W)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def kernel_timing(model, fl, B, dump=""):
    """Per-launch durations with HIP events on the launch stream (torch's current stream is the one every kernel of the
    program is enqueued on); aggregated per kernel class.  Returns the roofline object for the dominant kernel."""
    plan = next(reversed(model._plans.values()))
    P = plan.prog
    n = len(P)
    reps = 3
    tot = {}
    for _ in range(reps):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record()
        for i in range(n):
            P.run(i, i + 1)
            evs[i + 1].record()
        torch.cuda.synchronize()
        for i in range(n):
            cls, tag, flops, nbytes = P.meta[i]
            d = tot.setdefault(cls, {"ms": 0.0, "flops": 0.0, "launches": 0, "bytes": 0.0})
            d["ms"] += evs[i].elapsed_time(evs[i + 1])
            d["flops"] += flops
            d["bytes"] += nbytes
            d["launches"] += 1
            if tag.startswith("enc."):
                e = tot.setdefault(tag, {"ms": 0.0, "flops": 0.0, "launches": 0, "bytes": 0.0})
                e["ms"] += evs[i].elapsed_time(evs[i + 1]); e["flops"] += flops; e["launches"] += 1
    if dump:
        with open(dump, "w") as f:
            for i in range(n):
                cls, tag, flops, nbytes = P.meta[i]
                us = evs[i].elapsed_time(evs[i + 1]) * 1e3
                f.write(f"{i}\t{cls}\t{tag}\t{us:.1f}\t{flops / us / 1e6 if flops else 0:.1f}\n")
    classes = {k: v for k, v in tot.items() if not k.startswith("enc.")}
    dom = max(classes, key=lambda k: classes[k]["ms"])
    d = classes[dom]
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["flops"] else 0.0
    enc_ms = sum(tot[t]["ms"] for t in tot if t.startswith("enc.")) / reps
    enc_fl = sum(tot[t]["flops"] for t in tot if t.startswith("enc.")) / reps
    breakdown = {k: {"ms_per_step": round(v["ms"] / reps, 4), "launches": v["launches"] // reps,
                     "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None,
                     "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["bytes"] else None}
                 for k, v in sorted(tot.items(), key=lambda kv: -kv[1]["ms"])}
    # HBM bytes per launch of that kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, see the json)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    if os.path.exists(tpath):
        short = dom.replace("Cfg<", "(anonymous namespace)::Cfg<")
        for name, rec in json.load(open(tpath))["kernels"].items():
            if short in name:
                traffic = rec["hbm_total_bytes"]
    return {
        "roofline": {"bound": "mfma", "kernel": dom + " (v_mfma_f32_16x16x32_f16)",
                     "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                     "flop_per_launch": round(d["flops"] / d["launches"], 1),
                     "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 3),
                     "algorithmic_bytes_per_launch": None},
        "roofline_enc_attention_mlp": {"achieved": round(enc_fl / (enc_ms * 1e-3) / 1e12, 2), "peak": MFMA_PEAK_TFLOPS,
                                       "unit": "TFLOP/s", "frac": round(enc_fl / (enc_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                       "ms_per_step": round(enc_ms, 4)},
        "kernel_breakdown": breakdown,
    }


def cpu_baseline(cfg, sd, H, W):
    """CPU oracle (port of the reference's fp32 CPU path) on this host: bounded sample = one bs=8 pass of the same workload."""
    from oracle import restate
    n = min(os.cpu_count() or 1, int(os.environ.get("UD_CPU_BASELINE_THREADS", "32")))   # >32 threads oversubscribes this op mix
    torch.set_num_threads(n)
    orc = restate.OracleV2(cfg, sd)
    nimg = 8
    x = torch.randint(0, 256, (nimg, 3, H, W), dtype=torch.uint8, generator=torch.Generator().manual_seed(7))
    orc.infer(x[:1])                                   # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    orc.infer(x)
    dt = time.perf_counter() - t0
    return {"value": round(nimg / dt, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/restate.py fp32, {H}x{W}, bs={nimg} (the bench workload), 1 timed pass after a bs=1 warm-up ({dt:.1f} s)"}


if __name__ == "__main__":
    main()
