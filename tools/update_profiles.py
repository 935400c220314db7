#!/usr/bin/env python
"""Copy the judged summaries of a tools/profile_bench.sh run from gpurun_out/prof_<tag>/ into profiles/ (tracked):
kernel stats CSV (rocprofv3 --kernel-trace --stats) and per-kernel HBM traffic JSON (PMC FETCH_SIZE / WRITE_SIZE passes)."""
import collections, csv, json, shutil, sys

tag, name = sys.argv[1], sys.argv[2]
src = f"gpurun_out/prof_{tag}"
shutil.copy(f"{src}/trace/t_kernel_stats.csv", f"profiles/{name}_kernel_stats.csv")


def agg(path):
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        a[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return a


f, w = agg(f"{src}/fetch/f_counter_collection.csv"), agg(f"{src}/write/w_counter_collection.csv")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_bench.sh) over `bench.py --steps 3 --warmup 1`; "
                "per-launch averages in bytes. hbm_read = FETCH_SIZE(KB) * 1024 * 2 (gfx950 correction for wide coalesced reads, "
                "MI355X_MICROARCH.md section HBM; confirmed here on layernorm_kernel whose algorithmic read is 4*rows*D bytes), "
                "hbm_write = WRITE_SIZE(KB) * 1024 (uncorrected). Infinity-Cache hits are included in FETCH_SIZE.", "kernels": {}}
for k in f:
    fr = sum(f[k]) / len(f[k])
    wr = sum(w.get(k, [0])) / max(1, len(w.get(k, [0])))
    out["kernels"][k] = {"launches": len(f[k]), "hbm_read_bytes": round(fr * 2048), "hbm_write_bytes": round(wr * 1024),
                         "hbm_total_bytes": round(fr * 2048 + wr * 1024)}
json.dump(out, open(f"profiles/{tag}_hbm_traffic.json", "w"), indent=1)
import os
if os.path.exists(f"{src}/v1trace/v_kernel_stats.csv"):
    shutil.copy(f"{src}/v1trace/v_kernel_stats.csv", f"profiles/{tag}_v1_cnvnxtl_640x480_bs16_kernel_stats.csv")
if os.path.exists(f"gpurun_out/prof_{tag}.v1.log"):
    shutil.copy(f"gpurun_out/prof_{tag}.v1.log", f"profiles/{tag}_v1_cnvnxtl_640x480_bs16_breakdown.txt")
print("updated profiles/ from", src)
