#!/usr/bin/env python
"""Isolated vs in-situ cost of the encoder block's launches, in ONE process on one box.

The headline step spends ~45 us per encoder block more than the sum of the same launches benchmarked alone.  This probe times the
five launches of a block (qkv, attention, proj, fc1, fc2 of the ViT-L/14 518x518 bs=8 plan, taken from the real program) as
  seq        the whole encoder replayed as the step does it (one event pair around all 24 blocks; per-block average; no per-launch events)
  seq_ev     the same with an event pair around every launch (bench.py's kernel_timing method: what the per-launch tables report)
  warm       each launch alone, 30 x back to back (operands L2 / Infinity-Cache warm: what the isolated benchmarks measured)
  rotate     the launches of one kind from all 24 blocks back to back (weights: 24 distinct sets = L2-cold, Infinity-Cache-warm at best)
  cold       a 2 x 512 MB copy between launches (operands come from HBM), event pair around the launch only
and prints one table.  Under `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE` (or FETCH_SIZE) the same run gives clock and fetched
bytes per phase: --phases-json writes the launch counts per phase so tools/r4_insitu_post.py can slice the kernel trace.
GPU box only; mutates the plan's activations (accumulating launches are replayed many times) -- timing only."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--phases-json", default="")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--quick", action="store_true", help="fewer repetitions (profiled runs)")
    args = ap.parse_args()
    import torch
    from oracle import synth
    from unidepth_amd import UniDepthV2
    import warnings
    warnings.simplefilter("ignore")
    dev = torch.device("cuda", 0)
    cfg = synth.load_config("vitl14")
    sd = synth.make_synthetic_checkpoint(cfg, 125)
    model = UniDepthV2(cfg).load_state_dict(sd).to(dev).eval()
    g = torch.Generator().manual_seed(1)
    rgb = torch.randint(0, 256, (8, 3, 518, 518), dtype=torch.uint8, generator=g).to(dev)
    for _ in range(3):
        model.infer(rgb)
    torch.cuda.synchronize()
    plan = next(reversed(model._plans.values()))
    P = plan.prog
    first, last = plan.enc_first, plan.enc_last
    kinds = ["enc.qkv", "enc.attn", "enc.proj", "enc.fc1", "enc.fc2"]
    idx = {k: [i for i in range(first, last) if P.meta[i][1] == k] for k in kinds}
    blk = 6                                           # a folded block in the middle of the encoder
    reps = 8 if args.quick else args.reps
    phases = []                                       # (name, engine launches) in execution order, for the trace post-processor

    def ev():
        return torch.cuda.Event(enable_timing=True)

    def timed(fn):
        a, b = ev(), ev()
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3                # us

    res = {}
    # ---- seq: the encoder as the step runs it
    n_seq = 3 if args.quick else 6
    P.run(first, last); torch.cuda.synchronize()
    t = timed(lambda: [P.run(first, last) for _ in range(n_seq)])
    phases.append(("seq", n_seq * (last - first)))
    res["seq_per_block_us"] = t / n_seq / 24
    # ---- seq_ev: event pair around every launch, per kind (median over blocks 1..23 and repetitions)
    per = {k: [] for k in kinds}
    for _ in range(2 if args.quick else 3):
        evs = [ev() for _ in range(last - first + 1)]
        evs[0].record()
        for i in range(first, last):
            P.run(i, i + 1); evs[i - first + 1].record()
        torch.cuda.synchronize()
        for k in kinds:
            per[k] += [evs[i - first].elapsed_time(evs[i - first + 1]) * 1e3 for i in idx[k][1:]]
    phases.append(("seq_ev", (2 if args.quick else 3) * (last - first)))
    med = lambda v: sorted(v)[len(v) // 2]
    res["seq_ev"] = {k: med(per[k]) for k in kinds}
    # ---- warm: one launch repeated
    res["warm"] = {}
    for k in kinds:
        i = idx[k][blk]
        P.run(i, i + 1); torch.cuda.synchronize()
        res["warm"][k] = timed(lambda: [P.run(i, i + 1) for _ in range(reps)]) / reps
        phases.append(("warm." + k, reps + 1))
    # ---- rotate: the 24 launches of a kind back to back
    res["rotate"] = {}
    for k in kinds:
        ii = idx[k][1:]
        [P.run(i, i + 1) for i in ii]; torch.cuda.synchronize()
        nrot = 1 if args.quick else 2
        res["rotate"][k] = timed(lambda: [[P.run(i, i + 1) for i in ii] for _ in range(nrot)]) / (nrot * len(ii))
        phases.append(("rotate." + k, (nrot + 1) * len(ii)))
    # ---- warm_ev / cold: event pair around the single launch, with and without a cache flush before it
    fl_a = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    fl_b = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    res["warm_ev"], res["cold"] = {}, {}
    ncold = 4 if args.quick else 10
    for k in kinds:
        i = idx[k][blk]
        ws, cs = [], []
        for _ in range(ncold):
            a, b = ev(), ev()
            P.run(i, i + 1)
            a.record(); P.run(i, i + 1); b.record(); torch.cuda.synchronize()
            ws.append(a.elapsed_time(b) * 1e3)
        phases.append(("warm_ev." + k, 2 * ncold))
        for _ in range(ncold):
            a, b = ev(), ev()
            fl_b.copy_(fl_a)
            a.record(); P.run(i, i + 1); b.record(); torch.cuda.synchronize()
            cs.append(a.elapsed_time(b) * 1e3)
        phases.append(("cold." + k, ncold))
        res["warm_ev"][k], res["cold"][k] = med(ws), med(cs)
    # ---- report
    print(f"{'launch':10s} {'seq_ev':>8s} {'warm':>8s} {'rotate':>8s} {'warm_ev':>8s} {'cold':>8s}   (us; *_ev and cold include one event pair)")
    tot = {c: 0.0 for c in ("seq_ev", "warm", "rotate", "warm_ev", "cold")}
    for k in kinds:
        row = [res[c][k] for c in ("seq_ev", "warm", "rotate", "warm_ev", "cold")]
        for c, v in zip(tot, row):
            tot[c] += v
        print(f"{k:10s} " + " ".join(f"{v:8.1f}" for v in row))
    print(f"{'sum':10s} " + " ".join(f"{tot[c]:8.1f}" for c in tot))
    print(f"seq (no per-launch events): {res['seq_per_block_us']:.1f} us per block (incl. the block-0 / output LayerNorm launches, ~1.5 us per block)")
    res["sum"] = tot
    print("JSON " + json.dumps(res))
    if args.phases_json:
        json.dump({"phases": phases, "kinds": kinds}, open(args.phases_json, "w"))


if __name__ == "__main__":
    main()
