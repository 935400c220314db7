#!/usr/bin/env python
"""bench.py -- throughput of the UniDepthV2 infer() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 and no torch.distributed environment: bench.py re-launches itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU, RCCL); when the
driver already launched it that way (RANK / WORLD_SIZE set) it checks that the world size equals N.  The JSON line's
`n_gpus` is `dist.get_world_size()` -- never the flag.

Workload (BASELINE.json metric / configs[1]): UniDepthV2 ViT-L/14, 518x518, batch 8 per GPU, synthetic uint8 RGB already
resident in HBM, seeded random-init ("sensitised") weights of the exact architecture; a step = one full infer()
(pre-process -> 24-block encoder -> decoder -> all 7 outputs on device).  With N GPUs every rank runs its own batch
(weak scaling, images are independent) and the per-step outputs `depth`, `confidence`, `intrinsics` are all-gathered
over RCCL (one packed collective per step, issued on the step's own HIP stream so it overlaps the next step's compute);
value = N * 8 images * K / max-over-ranks wall time.

One JSON line on rank 0 with the driver contract fields plus
  value_one_call / p50_latency_ms -- the same workload with ONE infer() in flight (the headline `value` keeps `--inflight`
                  (default 2) independent calls in flight per GPU: throughput mode for a stream of requests);
  roofline     -- the dominant kernel class of the step (an MFMA GEMM instantiation, named as rocprofv3 prints it): algorithmic
                  FLOP per launch / average launch duration measured live with HIP events on the launch stream, against the
                  2.5 PFLOP/s dense fp16 peak; algorithmic bytes per launch (operands + outputs once); `traffic` = HBM bytes per
                  launch from the committed PMC passes (profiles/); `attainable_this_box` = a pure MFMA stream on random fp16 operands
                  (csrc/calib.hip) run in this process before the timed region, `frac_of_attainable` against it: comparable across boxes;
  rccl         -- N > 1: ranks seen, bytes gathered per rank and step, the collective's own duration, and how much of it the
                  timed region hides behind compute;
  cpu_baseline -- the CPU oracle (fp32 restatement of the reference, oracle/restate.py) timed on this box's host cores on a
                  bounded sample (one bs=8 pass of the same workload at 32 / 64 / all physical threads, then 2 more at the best thread
                  count: p50 of those 3; + 5 passes of BASELINE configs[0]), rank 0 / N=1 only.
  configs      -- N = 1 only, after the timed region (never part of `value`): the other BASELINE.json configurations that fit one GPU,
                  each a few seconds of GPU time: `v1_cnvnxtl_640x480_bs16` (configs[3]; own roofline, CPU baseline on bs=16, where fp32-class
                  arithmetic is used), `mixed_644x966+518x518_bs32` (configs[4] on one GPU through dist.infer_mixed), `knn_307200`
                  (the reference's native K-NN extension at the size its 3-D metrics use), `reference_as_shipped_rocm` (the reference's module
                  graph as PyTorch-ROCm ops under torch.autocast(fp16) on this same GPU and workload: the same-box comparator of value_one_call /
                  p50_latency_ms).  --no-extra-configs skips them.
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0      # dense fp16/bf16, /opt/skills/guides/MI355X_MICROARCH.md
# what the matrix pipes SUSTAIN on random fp16 operands (power / clock limited): the MFMA-only instruction stream of tools/ubench/gemm4w,
# 1462 TFLOP/s with random operands vs 2118 with constant ones (profiles/r02_mfma_attainable.txt).  Reported beside the datasheet fraction.
MFMA_ATTAINABLE_TFLOPS = 1460.0
HBM_PEAK_GBS = 8000.0


def flops_per_image(D=1024, depth=24, N=1370, C=512, hw=1369):
    """SURVEY.md 8(d) algorithmic work model (2 FLOP/MAC; softmax/LN/GELU not counted) for ViT-L/14 @ 518^2."""
    enc_gemm = depth * 2 * N * D * 12 * D
    enc_attn = depth * 4 * N * N * D
    return dict(enc_gemm=enc_gemm, enc_attn=enc_attn, enc=enc_gemm + enc_attn, total=1366.8e9)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)          # >= 30 timed iterations, >= 5 warm-up
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--arch", default="vitl14")
    ap.add_argument("--size", type=int, nargs=2, default=[518, 518])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the sub-records of the other BASELINE configs (V1, mixed list, K-NN)")
    ap.add_argument("--gather", default="depth,confidence,intrinsics")
    ap.add_argument("--gather-algo", default="collective", choices=["collective", "direct"],
                    help="exchange step: RCCL all_gather_into_tensor, or the all-pairs send/recv group (unidepth_amd.dist.all_gather_direct)")
    ap.add_argument("--dump-ops", default="", help="write per-launch timings (tsv) to this file")
    ap.add_argument("--inflight", type=int, default=2, help="infer() calls in flight per GPU during the timed steps (1 = one call at a time)")
    return ap.parse_args(argv)


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args) -> int:
    """`python bench.py --gpus N` outside torch.distributed: become the launcher of N ranks of this very script."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and not os.environ.get("UD_BENCH_SHARE_GPU"):
        print(json.dumps({"error": f"--gpus {args.gpus} requested but only {have} GPU(s) visible", "n_gpus": have}))
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("UD_BENCH_BACKEND", "nccl")      # "gloo": functional check of the N > 1 path on a box with one GPU
    if os.environ.get("UD_BENCH_SHARE_GPU"):                 # (all ranks on cuda:0; never a measurement)
        local_rank = 0
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            # RCCL's own account of what it built (transport per channel, algorithm / protocol per collective) goes to a per-rank file:
            # parsed into the `rccl` block so a scaling line carries the evidence that the exchange ran over xGMI P2P
            os.environ.setdefault("NCCL_DEBUG", "INFO")
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,GRAPH,COLL,P2P")
            os.environ.setdefault("NCCL_DEBUG_FILE", f"/tmp/ud_bench_rccl_{os.getpid()}_%h_%p.log")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))      # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s)")
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

    # N > 1 over RCCL: the exchange step runs through the C-ABI entry (ud_rccl_allgather_outputs, the library's own communicator; torch.distributed
    # only carries its unique id) when EVERY rank could create that communicator, otherwise through torch.distributed's RCCL group.  Which one ran
    # is reported in the `rccl` block.
    exchange_route = {"cabi": False, "note": "torch.distributed"}
    if world > 1 and backend == "nccl" and not os.environ.get("UD_BENCH_TORCH_EXCHANGE"):
        from unidepth_amd import dist as ud_dist
        try:
            ud_dist.init_cabi_exchange()
            mine_ok, why = 1, ""
        except Exception as e:                                   # e.g. librccl missing: every rank must take the same route
            mine_ok, why = 0, repr(e)[:200]
        flag = torch.tensor([mine_ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()):
            exchange_route.update(cabi=True, note="C-ABI ud_rccl_allgather_outputs (library communicator; torch.distributed = bootstrap only)")
        else:
            if mine_ok:
                ud_dist.finalize_cabi_exchange()
            exchange_route["note"] = "torch.distributed (C-ABI communicator not created on every rank" + (": " + why if why else "") + ")"

    def gather(out, algo=None):
        # one exchange step: requested outputs packed per image, ONE message per peer (xGMI is point-to-point: few, larger messages);
        # "collective" = RCCL all_gather_into_tensor, "direct" = all-pairs send / recv group (unidepth_amd/dist.py explains the choice)
        from unidepth_amd.dist import _cabi_allgather, all_gather_direct
        packed = torch.cat([out[k].reshape(B, -1) for k in gather_keys], dim=1)
        gathered = torch.empty((world * B, packed.shape[1]), dtype=packed.dtype, device=dev)
        if exchange_route["cabi"]:                                   # the library's own communicator (ud_rccl_allgather_outputs)
            _cabi_allgather(gathered, packed.contiguous(), (algo or args.gather_algo) == "direct")
        elif (algo or args.gather_algo) == "direct":
            all_gather_direct(gathered, packed)
        elif backend == "nccl":
            dist.all_gather_into_tensor(gathered, packed)
        else:
            dist.all_gather(list(gathered.chunk(world)), packed)
        return gathered

    def step(with_gather=True):
        """One full infer() of the batch; up to --inflight steps overlap on separate HIP streams (independent batches of a
        stream of requests).  The all-gather of a step is issued on that step's stream, right behind its outputs."""
        return pipe.submit(rgb, post=gather if (world > 1 and with_gather) else None)

    def step_single():
        out = model.infer(rgb)
        if world > 1:
            gather(out)
        return out

    def timed(n, with_gather=True):
        pipe.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step(with_gather)
        pipe.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = tt.item()
        return dt

    for _ in range(args.warmup):
        step()
    calib = mfma_attainable(torch, dev) if (rank == 0 and not args.no_kernel_timing) else None     # after the warm-up (clocks up), before the timed region
    for _ in range(2):
        step()
    elapsed = timed(args.steps)                      # THE timed region: exactly K steps, barrier + synchronize on both sides, max over ranks
    # latency of ONE call with nothing else in flight (outside the timed region)
    lat = []
    for _ in range(max(30, min(args.steps, 50))):
        ts = time.perf_counter()
        step_single()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - ts) * 1e3)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    p50 = statistics.median(lat)
    p90 = sorted(lat)[int(0.9 * (len(lat) - 1) + 0.5)]

    result = {
        "metric": "images/sec (whole node) + p50 latency, ViT-L/14 518x518 bs=8",
        "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "p50_latency_ms": round(p50, 4), "p90_latency_ms": round(p90, 4), "latency_samples": len(lat),
        "value_one_call": round(world * B / (p50 * 1e-3), 3),
        "inflight": max(1, args.inflight),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
        "data": "synthetic uint8 RGB (seeded) resident in HBM; seeded random-init weights of the named architecture",
        "config": {"workload": f"UniDepthV2 {args.arch} infer(), {H}x{W}, bs={B} per GPU, all 7 outputs on device; "
                               f"value = {max(1, args.inflight)} independent infer() calls in flight per GPU (HIP streams); "
                               f"value_one_call / p50_latency_ms = one call at a time",
                   "global_batch": world * B, "parallelism": f"dp{world}" + (f" + RCCL all-gather({','.join(gather_keys)})" if world > 1 else "")},
    }

    if world > 1:
        result["rccl"] = rccl_report(torch, dist, model, rgb, gather, timed, args, world, rank, dev, backend, B, ms_per_step)
        result["rccl"]["exchange_route"] = exchange_route["note"]
    if rank == 0:
        fl = flops_per_image()
        result["model_tflops_per_s"] = round(value * fl["total"] / 1e12 / world, 2)
        # the metric is "images/sec + p50 latency": both halves also inside `config` / `roofline`, the objects every consumer of the line keeps
        result["config"].update({"value_one_call": result["value_one_call"], "p50_latency_ms": result["p50_latency_ms"],
                                 "p90_latency_ms": result["p90_latency_ms"], "inflight": result["inflight"]})
        if not args.no_kernel_timing:
            result.update(kernel_timing(torch, model, fl, B, args.dump_ops, calib))
            scope = result["roofline_enc_attention_mlp"]
            result["roofline"].update({"scope_frac": scope["frac"], "scope_achieved": scope["achieved"], "scope_ms_per_step": scope["ms_per_step"],
                                       "scope": "encoder attention + MLP blocks (north_star target 0.60): see roofline_enc_attention_mlp",
                                       "one_call_p50_ms": result["p50_latency_ms"], "one_call_images_per_s": result["value_one_call"]})
        if world == 1 and not args.no_extra_configs and args.arch == "vitl14" and tuple(args.size) == (518, 518):
            del pipe
            model.clear_plans()
            result["configs"] = extra_configs(torch, model, dev, cpu=not args.no_cpu_baseline)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(torch, cfg, sd, H, W)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        if exchange_route["cabi"]:
            ud_dist.finalize_cabi_exchange()
        dist.destroy_process_group()


def rccl_report(torch, dist, model, rgb, gather, timed, args, world, rank, dev, backend, B, ms_with):
    """Evidence that the exchange step ran over RCCL on `world` ranks, what it moved, what it cost alone and inside the pipeline."""
    ranks = torch.empty(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([rank], dtype=torch.int64, device=dev)
    if backend == "nccl":
        dist.all_gather_into_tensor(ranks, mine)
    else:
        dist.all_gather(list(ranks.chunk(world)), mine)
    out = model.infer(rgb)
    torch.cuda.synchronize()
    g = gather(out)
    torch.cuda.synchronize()
    ok = True
    for r in range(world):                                   # every rank's block of the gathered outputs must be finite and non-zero
        blk = g[r * B:(r + 1) * B]
        ok = ok and bool(torch.isfinite(blk).all()) and bool((blk.abs().sum(dim=1) > 0).all())
    reps = 10
    alone = {}
    # both forms of the exchange step, timed alone on the same outputs (gloo, the 1-GPU functional stand-in, has no GPU send / recv)
    for algo in (("collective", "direct") if backend == "nccl" else (args.gather_algo,)):
        ga = gather(out, algo)
        torch.cuda.synchronize()
        same = bool(torch.equal(ga, g))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(reps):
            gather(out, algo)
        ev1.record()
        torch.cuda.synchronize()
        alone[algo] = (ev0.elapsed_time(ev1) / reps, same)
    gather_ms = alone[args.gather_algo][0]
    n2 = max(4, args.steps // 2)
    ms_without = timed(n2, with_gather=False) / n2 * 1e3
    exposed = max(0.0, ms_with - ms_without)
    per_rank = int(g.shape[1]) * B * g.element_size()
    # where the ranks sit: PCI bus id of every rank's device (distinct ids = distinct GPUs) + what RCCL logged about its transports
    props = torch.cuda.get_device_properties(dev)
    bus = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))
    buses = [None] * world
    dist.all_gather_object(buses, bus)
    return {"backend": "rccl (torch.distributed 'nccl')" if backend == "nccl" else backend, "world_size": world,
            "ranks_seen": [int(x) for x in ranks.tolist()], "all_blocks_valid": ok,
            "pci_bus_ids": buses, "distinct_devices": len(set(buses)),
            "gather_algo": args.gather_algo,
            "gather_alone_ms_by_algo": {k: round(v[0], 4) for k, v in alone.items()},
            "algos_bit_identical": all(v[1] for v in alone.values()),
            "xgmi_model": {"links_per_gpu": 7, "GBps_per_link": 153.0,
                           "direct_ms": round(per_rank / 153e9 * 1e3, 4), "ring_ms": round((world - 1) * per_rank / 153e9 * 1e3, 4),
                           "note": "full-mesh xGMI: a ring all-gather is per-link bound over world-1 hops, the all-pairs form uses every link at once "
                                   "(SURVEY.md 8e); model only -- compare with gather_alone_ms_by_algo"},
            "rccl_log": rccl_log_summary() if backend == "nccl" else None,
            "gathered_bytes_per_rank_per_step": per_rank, "gather_alone_ms": round(gather_ms, 4),
            "gather_alone_GBps_per_rank_in": round(per_rank * (world - 1) / (gather_ms * 1e-3) / 1e9, 2),
            "ms_per_step_without_gather": round(ms_without, 4), "exposed_gather_ms_per_step": round(exposed, 4),
            "overlap_frac": round(1.0 - min(1.0, exposed / gather_ms), 3) if gather_ms > 0 else None}


def rccl_log_summary():
    """What RCCL itself logged while building the communicator and running the collectives (NCCL_DEBUG=INFO into NCCL_DEBUG_FILE):
    transports per channel (P2P/IPC = xGMI or PCIe peer access, SHM, NET), ring / tree graphs, algorithm + protocol per collective."""
    import glob
    import re
    pat = os.environ.get("NCCL_DEBUG_FILE", "")
    if not pat:
        return None
    files = glob.glob(re.sub(r"%[hp]", "*", pat))
    text = ""
    for f in files[:1]:
        try:
            text = open(f, errors="replace").read()
        except OSError:
            pass
    if not text:
        return {"file": None}
    via = sorted(set(re.findall(r"via (P2P/[A-Za-z/]+|SHM[/A-Za-z]*|NET/[A-Za-z0-9_/]+|direct shared memory)", text)))
    algos = sorted(set(re.findall(r"(?:algo(?:rithm)?\s*[=:]?\s*)(\d|Ring|Tree|Direct|CollNet\w*|NVLS\w*)", text, flags=re.I)))
    protos = sorted(set(re.findall(r"(?:proto(?:col)?\s*[=:]?\s*)(\d|LL128|LL|Simple)", text, flags=re.I)))
    return {"transports": via, "algo_tokens": algos, "proto_tokens": protos,
            "xgmi_mentions": len(re.findall(r"xgmi|XGMI", text)), "channels": len(set(re.findall(r"Channel (\d+)", text))),
            "rccl_version": (re.findall(r"(?:RCCL|NCCL) version ([0-9.+\-a-zA-Z]+)", text) or [None])[0],
            "lines": text.count("\n")}


def mfma_attainable(torch, dev):
    """What the matrix pipes of THIS box sustain on a pure MFMA stream with random fp16 operands (csrc/calib.hip, ~0.3 s): boxes of the pool
    differ by several per cent on every kernel, so fractions against a per-box number are comparable across boxes where the datasheet
    fraction is not.  Runs before the timed region; HIP events on the launch stream."""
    import ctypes as C
    from unidepth_amd import _lib
    ops_t = (torch.randn(1 << 20, generator=torch.Generator().manual_seed(11)) * 0.5).half().to(dev)       # 2 MiB
    sink = torch.zeros(1024 * 512, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    # round 6: two streams -- the 32 x 32 x 16 one of rounds 2-5 and the GEMM family's own 16 x 16 x 32, which sustains more on random operands
    # (profiles/r06_kloop_ablation.txt); the higher one is the box's attainable rate
    streams = (("ud_calib_mfma_stream", 1024, 640, "v_mfma_f32_32x32x16_f16 only, 16 independent per iteration and wave, 2 waves per SIMD"),
               ("ud_calib_mfma_stream16", 256, 2560, "v_mfma_f32_16x16x32_f16 only, 16 independent per iteration and wave, 2 waves per SIMD"))
    res = {}
    for fn_name, wgs, iters, what in streams:
        fn = getattr(_lib.lib, fn_name)
        flop = C.c_double(0.0)

        def launch():
            _lib.check(fn(ops_t.data_ptr(), iters, wgs, sink.data_ptr(), C.byref(flop), st), fn_name)
        for _ in range(20):
            launch()
        torch.cuda.synchronize()
        best, tot, n = 0.0, 0.0, 0
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                launch()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 50
            tf = flop.value / (ms * 1e-3) / 1e12
            best = max(best, tf); tot += tf; n += 1
        res[fn_name] = {"tflops": round(tot / n, 1), "best_tflops": round(best, 1), "launch_ms": round(flop.value / (tot / n) / 1e9, 4),
                        "stream": what + ", random fp16 operands (csrc/calib.hip)"}
    assert float(sink.abs().sum()) == 0.0
    top = max(res.values(), key=lambda r: r["tflops"])
    return dict(top, per_stream={k.replace("ud_calib_mfma_", ""): v["tflops"] for k, v in res.items()})


def kernel_timing(torch, model, fl, B, dump="", attainable=None):
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
            if tag.startswith("enc."):      # encoder block scope: qkv / attention / proj / fc1 / fc2 AND the two LayerNorms of Block.forward
                e = tot.setdefault(tag, {"ms": 0.0, "flops": 0.0, "launches": 0, "bytes": 0.0})
                e["ms"] += evs[i].elapsed_time(evs[i + 1]); e["flops"] += flops; e["launches"] += 1; e["bytes"] += nbytes
    if dump:
        with open(dump, "w") as f:
            for i in range(n):
                cls, tag, flops, nbytes = P.meta[i]
                us = evs[i].elapsed_time(evs[i + 1]) * 1e3
                f.write(f"{i}\t{cls}\t{tag}\t{us:.1f}\t{flops / us / 1e6 if flops else 0:.1f}\t{nbytes / us / 1e3 if nbytes else 0:.1f}\n")
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
    traffic, traffic_src = None, None
    for name in ("r06_hbm_traffic.json", "r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01_hbm_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        if traffic is None and os.path.exists(tpath):
            short = dom.replace("Cfg<", "(anonymous namespace)::Cfg<")
            for kname, rec in json.load(open(tpath))["kernels"].items():
                if short in kname:
                    traffic, traffic_src = rec["hbm_total_bytes"], "profiles/" + name
    if attainable:
        att, att_src = attainable["tflops"], ("measured in this run before the timed region: " + attainable["stream"] +
                                              f"; best of 4 rounds {attainable['best_tflops']}; per stream {attainable.get('per_stream')}; rounds 2-5 quoted the 32x32x16 stream only "
                                              f"({MFMA_ATTAINABLE_TFLOPS} on the round-2 box, profiles/r02_mfma_attainable.txt)")
    else:
        att, att_src = MFMA_ATTAINABLE_TFLOPS, "profiles/r02_mfma_attainable.txt (constant from another box: no calibration in this run)"
    return {
        "roofline": {"bound": "mfma", "kernel": dom + " (v_mfma_f32_16x16x32_f16)",
                     "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "attainable_this_box": att, "frac_of_attainable": round(achieved / att, 4),
                     "attainable_source": att_src,
                     "flop_per_launch": round(d["flops"] / d["launches"], 1),
                     "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 3),
                     "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"], 1),
                     "note": ("a kernel CLASS (template instantiation), not one shape: in the ViT-L step this one holds the encoder's proj (K = 1024, 23 GFLOP) "
                              "and fc2 (K = 4096, 92 GFLOP) launches with the fp32 residual accumulate epilogue; where the LayerNorm fold is on "
                              "(DESIGN 9.2) the same launches also write the fp16 copy of their rows and the rows' LayerNorm statistics, i.e. "
                              "they carry the work of the 47 LayerNorm launches that are gone from the step -- the block-scope fraction below "
                              "is the one to compare across rounds") if dom.startswith(("gemm256_kernel<3, 1, 0", "gemm_pp_f32_kernel")) else None},
        "roofline_enc_attention_mlp": {"achieved": round(enc_fl / (enc_ms * 1e-3) / 1e12, 2), "peak": MFMA_PEAK_TFLOPS,
                                       "unit": "TFLOP/s", "frac": round(enc_fl / (enc_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                       "frac_of_attainable": round(enc_fl / (enc_ms * 1e-3) / 1e12 / att, 4),
                                       "ms_per_step": round(enc_ms, 4),
                                       "scope": "the 24 encoder blocks as the reference's Block.forward runs them: both LayerNorms (or what is left of them "
                                                "after folding), qkv, attention, proj, fc1 + GELU, fc2, LayerScale and residual adds; "
                                                "FLOP = GEMMs + QK^T / PV only (SURVEY.md 8d)",
                                       "launches_per_step": sum(tot[t]["launches"] for t in tot if t.startswith("enc.")) // reps},
        "kernel_breakdown": breakdown,
    }


def _timed_calls(torch, fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


def _program_breakdown(torch, P, reps=2):
    """HIP events around every launch of a recorded program: {kernel class: [ms, flops, launches]} per replay."""
    n = len(P)
    tot = {}
    for rep in range(reps + 1):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record()
        for i in range(n):
            P.run(i, i + 1)
            evs[i + 1].record()
        torch.cuda.synchronize()
        if rep:
            for i in range(n):
                cls, tag, fl, nb = P.meta[i]
                d = tot.setdefault(cls, [0.0, 0.0, 0])
                d[0] += evs[i].elapsed_time(evs[i + 1]) / reps; d[1] += fl / reps; d[2] += 1
    return {k: [v[0], v[1], v[2] // reps] for k, v in tot.items()}


def latency_bs1(torch, model_vitl, dev):
    """p50 wall time of ONE bs=1 infer() (all 7 outputs on the device, synchronised per call).  The program of such a call is ~240 launches of
    2-20 us, paced on the GPU by dependent, under-filled kernels (hipGraph replay of the same program measured 0.00 ms faster in round 4)."""
    import statistics
    from oracle import synth
    from unidepth_amd import UniDepthV2
    rec = {"metric": "p50 latency of one bs=1 infer() call", "unit": "ms", "higher_is_better": False, "calls": 30}
    cfg_s = synth.load_config("vits14")
    m_s = UniDepthV2(cfg_s).load_state_dict(synth.make_synthetic_checkpoint(cfg_s, 7)).to(dev).eval()
    m_s.resolution_level = 2
    for tag, m, (H, W) in (("vitl14_518x518", model_vitl, (518, 518)), ("vits14_462x616", m_s, (462, 616))):
        rgb = torch.randint(0, 256, (1, 3, H, W), dtype=torch.uint8, generator=torch.Generator().manual_seed(3)).to(dev)
        m.clear_plans()
        for _ in range(3):
            m.infer(rgb)
        torch.cuda.synchronize()
        ts = []
        for _ in range(rec["calls"]):
            t0 = time.perf_counter()
            m.infer(rgb)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        plan = next(reversed(m._plans.values()))
        rec[tag] = {"p50_ms": round(statistics.median(ts), 3), "min_ms": round(min(ts), 3), "launches": len(plan.prog)}
        m.clear_plans()
    del m_s
    return rec

def reference_as_shipped(torch, dev, B=8, H=518, W=518, warm=5, calls=10):
    from oracle import restate, synth                    # baseline leg only (the checker, timed as the comparator)
    cfg = synth.load_config("vitl14")
    sd = synth.make_synthetic_checkpoint(cfg, 125)
    orc = restate.OracleV2(cfg, sd)
    orc.w = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in orc.w.items()}
    rgb = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).to(dev)

    def call():
        with torch.no_grad(), torch.device(dev), torch.autocast(device_type="cuda", enabled=True, dtype=torch.float16):
            return orc.infer(rgb)
    for _ in range(warm):
        call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter()
        call()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    p50 = statistics.median(ts)
    return {"metric": "images/sec + p50 latency, ViT-L/14 518x518 bs=8: the reference's module graph in PyTorch-ROCm under torch.autocast(fp16), one call at a time",
            "value": round(B / (p50 * 1e-3), 2), "unit": "images/s", "p50_latency_ms": round(p50, 3), "min_ms": round(min(ts), 3), "calls": calls, "warmup": warm,
            "kind": "port (oracle/restate.py ops on cuda; skips the dead work the live reference performs: SURVEY 8a-20)", "torch": torch.__version__,
            "compare_with": "value_one_call / p50_latency_ms of the headline line (same box, same inputs, one call at a time)"}


def extra_configs(torch, model_v2, dev, cpu=True):
    """The other BASELINE.json configurations that fit one GPU (configs[3], configs[4] on one GPU, the K-NN extension), each a few
    seconds of GPU time, after the timed region of the headline metric."""
    out = {}
    # ---- configs[3]: UniDepthV1 ConvNeXt-L, 640x480, bs=16
    try:
        from oracle import synth_v1
        from unidepth_amd import UniDepthV1
        from unidepth_amd import unidepthv1 as v1mod
        cfg1 = synth_v1.load_config_v1()
        sd1 = synth_v1.make_synthetic_checkpoint_v1(cfg1, 211)
        m1 = UniDepthV1(cfg1).load_state_dict(sd1).to(dev).eval()
        B1 = 16
        rgb1 = torch.randint(0, 256, (B1, 3, 480, 640), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).to(dev)
        dt = _timed_calls(torch, lambda: m1.infer(rgb1), 10, warm=3)
        plan = next(reversed(m1._plans.values()))
        tot = _program_breakdown(torch, plan.prog)
        dom, dv = max(((k, v) for k, v in tot.items() if v[1] > 0), key=lambda kv: kv[1][0])
        fl_total = sum(v[1] for v in tot.values())
        rec = {"metric": "images/sec, UniDepthV1 ConvNeXt-L 640x480 bs=16 (BASELINE configs[3])", "value": round(B1 / dt, 2), "unit": "images/s",
               "ms_per_step": round(dt * 1e3, 3), "steps": 10, "launches": len(plan.prog),
               "dtype": "f16 MFMA operands, fp32 accumulate" + ("; weights as TWO fp16 terms (W_hi + W_lo, ~22-bit weights: UdGemm.a_wrap)" + ("" if v1mod.WSPLIT_CONVNEXT_FC1 else " except the ConvNeXt blocks' fc1") if v1mod.WSPLIT else "")
                        + ("; the ConvUpsample tails and the output convs as THREE-term products ([A_hi | A_lo] x [W_hi | W_hi | W_lo])" if getattr(v1mod, "ASPLIT", False) else "")
                        + "; depth-wise convolutions, LayerNorm / softmax statistics, the camera transformer, the NystromBlocks' per-token head attention and all residual streams in fp32",
               "parity": "depth ARel <= 1e-3 per image vs the fp32 oracle at this batch (tests/test_v1_gpu.py::test_v1_infer_config4_bs16_vs_oracle); over 8 checkpoint "
                         "seeds x 2 sizes (tests/test_parity_sweep_gpu.py): median 5.9e-4, max 7.9e-4, 16 of 16 within 1e-3 (profiles/r05_parity_sweep.txt); "
                         "NystromBlocks as the reference executes them (oracle/stubs/xformers restates xformers' NystromAttention)",
               "model_tflops_per_s": round(fl_total / dt / 1e12, 1),
               "roofline": {"bound": "mfma", "kernel": dom + " (v_mfma_f32_16x16x32_f16)", "achieved": round(dv[1] / (dv[0] * 1e-3) / 1e12, 2),
                            "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(dv[1] / (dv[0] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                            "flop_per_launch": round(dv[1] / dv[2], 1), "avg_launch_us": round(dv[0] * 1e3 / dv[2], 2), "traffic": None,
                            "note": "algorithmic FLOP (2 M N K of the layer): the second fp16 term of the split weights is not counted as work"}}
        if cpu:
            from oracle import restate_v1                    # CPU baseline leg only: the checker, timed on the host cores
            model, phys, logical = host_cpu()
            torch.set_num_threads(min(phys, 32))
            orc = restate_v1.OracleV1(cfg1, sd1)
            x = rgb1.cpu()
            orc.infer(x[:1])
            t0 = time.perf_counter()
            for i in range(0, B1, 4):
                orc.infer(x[i:i + 4])
            t = time.perf_counter() - t0
            rec["cpu_baseline"] = {"value": round(B1 / t, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port", "seconds": round(t, 2),
                                   "sample": "oracle/restate_v1.py fp32, the 16 images of the bench batch (4 calls of 4), one pass after a 1-image warm-up"}
        out["v1_cnvnxtl_640x480_bs16"] = rec
        del m1, plan
    except Exception as e:                                   # a sub-record must never take the headline line down
        out["v1_cnvnxtl_640x480_bs16"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    # ---- configs[4] on ONE GPU: 16 x 644x966 + 16 x 518x518 through dist.infer_mixed
    try:
        from unidepth_amd.dist import infer_mixed
        g = torch.Generator().manual_seed(1)
        imgs = [torch.randint(0, 256, (3, 644, 966), dtype=torch.uint8, generator=g).to(dev) for _ in range(16)] + \
               [torch.randint(0, 256, (3, 518, 518), dtype=torch.uint8, generator=g).to(dev) for _ in range(16)]
        model_v2.max_plans = max(model_v2.max_plans, 12)
        dt = _timed_calls(torch, lambda: infer_mixed(model_v2, imgs, inflight=2), 3, warm=2)
        fl = 16 * 3708.1e9 + 16 * 1366.8e9                   # SURVEY.md 8d work model at the two network resolutions
        out["mixed_644x966+518x518_bs32"] = {"metric": "images/sec, UniDepthV2 ViT-L/14, 16 x 644x966 + 16 x 518x518 (BASELINE configs[4] on one GPU)",
                                             "value": round(32 / dt, 2), "unit": "images/s", "ms_per_list": round(dt * 1e3, 2), "steps": 3,
                                             "model_tflops_per_s": round(fl / dt / 1e12, 1), "frac_mfma_peak": round(fl / dt / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                             "dtype": "f16", "how": "dist.infer_mixed: bucketed by shape, micro-batches on 2 HIP streams"}
        model_v2.clear_plans()
    except Exception as e:
        out["mixed_644x966+518x518_bs32"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    # ---- the reference AS SHIPPED on this same GPU: the oracle's torch ops (oracle/restate.py = the reference's own module
    #      graph, pinned to it) on `cuda` under the autocast decorator of the reference's infer() (unidepthv2.py:239-240), i.e. PyTorch-ROCm
    #      (rocBLAS / hipBLASLt GEMMs, MIOpen convolutions, the SDPA kernel torch picks) on the bench workload.  A baseline leg like cpu_baseline:
    #      after the timed region, never part of `value`, nothing under unidepth_amd/ imports it.
    try:
        out["reference_as_shipped_rocm"] = reference_as_shipped(torch, dev)
    except Exception as e:
        out["reference_as_shipped_rocm"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    # ---- small-batch latency: bs = 1 p50 per infer()
    #      for ViT-L 518x518 and for BASELINE configs[0]'s shape (ViT-S, one 462x616 image)
    try:
        out["latency_bs1"] = latency_bs1(torch, model_v2, dev)
    except Exception as e:
        out["latency_bs1"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    # ---- the reference's K-NN extension at the size its 3-D metrics run on (one 480x640 depth map per cloud)
    try:
        from unidepth_amd import eval_ops
        g = torch.Generator().manual_seed(0)
        Pn = 480 * 640
        x = torch.randn(1, Pn, 3, generator=g)
        y = x[:, torch.randperm(Pn, generator=g)] + 0.01 * torch.randn(1, Pn, 3, generator=g)
        xd, yd = x.to(dev), y.to(dev)
        dt = _timed_calls(torch, lambda: eval_ops.knn_points(xd, yd, K=1), 5, warm=1)
        tf = Pn * Pn * 8 / dt / 1e12
        rec = {"metric": "K-NN (K=1, D=3) of 307200 points against 307200 points: ud_knn_points", "value": round(Pn * Pn / dt, 0), "unit": "point pairs/s",
               "ms_per_call": round(dt * 1e3, 3), "steps": 5, "dtype": "f32 (bit-exact with the reference's CPU build)",
               "roofline": {"bound": "valu_f32", "achieved": round(tf, 2), "peak": 78.6, "unit": "TFLOP/s (fp32 add/mul, no FMA)", "frac": round(tf / 78.6, 3)}}
        if cpu:
            from oracle import build_ref_knn                 # CPU baseline leg only: the reference's own CPU K-NN built from its sources
            ref = build_ref_knn.load_ref()
            if ref is not None:
                nq = 8192
                torch.set_num_threads(1)
                t0 = time.perf_counter()
                idx, d = ref.knn_points_idx(x[:, :nq].contiguous(), y, torch.tensor([nq]), torch.tensor([Pn]), 2, 1, -1)
                t = time.perf_counter() - t0
                r = eval_ops.knn_points(xd[:, :nq].contiguous(), yd, K=1)
                rec["cpu_baseline"] = {"value": round(nq * Pn / t, 0), "unit": "point pairs/s", "cores": 1, "kind": "reference", "seconds": round(t, 2),
                                       "sample": f"{nq} of {Pn} queries against all {Pn} points (oracle/_ref/knn/KNN.so)",
                                       "matches_gpu_bit_exact": bool(torch.equal(r.idx.cpu(), idx) and torch.equal(r.dists.cpu(), d))}
        out["knn_307200"] = rec
    except Exception as e:
        out["knn_307200"] = {"error": repr(e)}
    return out


def host_cpu():
    """CPU model and physical core count of this box (Linux /proc/cpuinfo)."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return model, (len(cores) or logical), logical


def cpu_baseline(torch, cfg, sd, H, W):
    """CPU oracle (port of the reference's fp32 CPU path, oracle/restate.py) on this host: >= 3 timed bs=8 passes of the bench workload
     + BASELINE configs[0] (ViT-S/14, 462x616, bs=1; 5 passes).  UD_CPU_BASELINE_REPS / _THREADS override."""
    from oracle import restate, synth
    model, phys, logical = host_cpu()
    cap = int(os.environ.get("UD_CPU_BASELINE_THREADS", "0"))
    reps = max(1, int(os.environ.get("UD_CPU_BASELINE_REPS", "3")))
    orc = restate.OracleV2(cfg, sd)
    nimg = 8
    x = torch.randint(0, 256, (nimg, 3, H, W), dtype=torch.uint8, generator=torch.Generator().manual_seed(7))
    # thread count: one timed pass at 32, 64 and all physical cores (SURVEY.md 8d asks for all cores; the op mix -- addmm, direct
    # convolutions, SDPA -- does not always scale that far), the best one is then run to `reps` passes and reported with its count
    sweep = {}
    cands = [cap] if cap > 0 else sorted({min(phys, c) for c in (32, 64, phys)})
    for n in cands:
        torch.set_num_threads(n)
        orc.infer(x[:1])                               # warm-up (thread pool, allocator)
        t0 = time.perf_counter()
        orc.infer(x)
        sweep[n] = time.perf_counter() - t0
    n = min(sweep, key=sweep.get)
    torch.set_num_threads(n)
    ts = [sweep[n]]
    for _ in range(reps - 1):
        t0 = time.perf_counter()
        orc.infer(x)
        ts.append(time.perf_counter() - t0)
    p50 = statistics.median(ts)
    # BASELINE configs[0]: the reference's own CPU-runnable case
    cfg_s = synth.load_config("vits14")
    orc_s = restate.OracleV2(cfg_s, synth.make_synthetic_checkpoint(cfg_s, 123))
    xs = torch.randint(0, 256, (1, 3, 462, 616), dtype=torch.uint8, generator=torch.Generator().manual_seed(1))
    orc_s.infer(xs)
    t1 = []
    for _ in range(5):
        t0 = time.perf_counter()
        orc_s.infer(xs)
        t1.append(time.perf_counter() - t0)
    return {"value": round(nimg / p50, 4), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "p50_s": round(p50, 3), "passes_s": [round(t, 3) for t in ts],
            "thread_sweep_s_per_pass": {str(k): round(v, 3) for k, v in sweep.items()},
            "cpu_model": model, "host_physical_cores": phys, "host_logical_cpus": logical,
            "sample": f"oracle/restate.py fp32, {H}x{W}, bs={nimg} (the bench workload): one pass per candidate thread count, then {reps} passes at the best "
                      f"({n} threads), p50",
            "config0_vits_462x616_bs1": {"value": round(1.0 / statistics.median(t1), 3), "unit": "images/s", "p50_s": round(statistics.median(t1), 4),
                                         "passes": 5},
            **_port_vs_reference(nimg / p50)}


def _port_vs_reference(port_value):
    """The `port` (oracle/restate.py) skips dead work the live reference performs.  Both were timed once on identical input and
    threads in the authoring container (tools/port_vs_reference.py -> profiles/r04_port_vs_reference.json; /root/reference does not exist on
    the GPU box): the ratio corrects the port number to what the reference itself would reach on these cores."""
    try:
        r = json.load(open(os.path.join(ROOT, "profiles", "r04_port_vs_reference.json")))
        ratio = float(r["port_vs_reference"])
        return {"port_vs_reference": ratio, "reference_equivalent_value": round(port_value / ratio, 4),
                "port_vs_reference_source": f"profiles/r04_port_vs_reference.json ({r['workload']}, {r['threads']} threads: port {r['port_s']} s, "
                                            f"live reference {r['reference_s']} s per call)"}
    except Exception:
        return {"port_vs_reference": None}


if __name__ == "__main__":
    main()
