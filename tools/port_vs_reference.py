#!/usr/bin/env python
"""Authoring container only (needs /root/reference): time the CPU oracle (oracle/restate.py, the `cpu_baseline.kind = "port"` of
bench.py) against the LIVE reference's own infer() on the same ViT-L/14 518x518 bs=1 input and the same thread count, so that the
"port" number can be corrected for the dead work the restatement skips (20 of 24 final LayerNorms, PositionEmbeddingSine, two
discarded resamples -- SURVEY 8a-20).  Writes profiles/r04_port_vs_reference.json; bench.py copies the ratio into cpu_baseline."""
import json, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_loader, restate, synth

assert ref_loader.available(), "needs /root/reference"
threads = int(os.environ.get("THREADS", str(os.cpu_count())))
torch.set_num_threads(threads)
cfg = synth.load_config("vitl14")
sd = synth.make_synthetic_checkpoint(cfg, 125)
g = torch.Generator().manual_seed(1)
rgb = torch.randint(0, 256, (1, 3, 518, 518), dtype=torch.uint8, generator=g)
orc = restate.OracleV2(cfg, sd)
ref = ref_loader.build_reference("vitl14", sd)


def timeit(fn, n=5):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts), ts


with torch.no_grad():
    a = orc.infer(rgb)
    b = ref.infer(rgb)
    err = ((a["depth"] - b["depth"]).abs() / b["depth"]).mean().item()
    t_port, all_port = timeit(lambda: orc.infer(rgb))
    t_ref, all_ref = timeit(lambda: ref.infer(rgb))
out = {"workload": "UniDepthV2 ViT-L/14 518x518 bs=1, fp32, CPU", "threads": threads, "host": os.uname().nodename, "cpu_count": os.cpu_count(),
       "port_s": round(t_port, 4), "reference_s": round(t_ref, 4), "port_vs_reference": round(t_ref / t_port, 4),
       "port_all_s": [round(x, 4) for x in all_port], "reference_all_s": [round(x, 4) for x in all_ref], "depth_arel_port_vs_reference": err,
       "note": "port_vs_reference = reference time / port time on identical input and threads: multiply a `port` images/s figure by 1 / this "
               "to estimate what the live reference would reach on the same cores"}
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_port_vs_reference.json"), "w"), indent=1)
