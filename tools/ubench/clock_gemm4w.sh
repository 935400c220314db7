#!/bin/bash
# average shader clock during the gemm4w variants: GRBM_GUI_ACTIVE (busy cycles) / kernel duration, random vs constant operands
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for mode in random const; do
  arg=""; [ $mode = const ] && arg="const"
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/clk_$mode -o c -- $R/tools/ubench/gemm4w 16384 4096 16384 $arg > $R/gpurun_out/clk_$mode.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for mode in ("random", "const"):
    f = glob.glob(f"/root/repo/gpurun_out/clk_{mode}/**/*counter_collection.csv", recursive=True)
    t = glob.glob(f"/root/repo/gpurun_out/clk_{mode}/**/*kernel_trace.csv", recursive=True)
    if not f or not t: print(mode, "no data"); continue
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(t[0]))}
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "gemm4w" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]) / max(1, dur.get(r["Dispatch_Id"], 1)))
    for k, v in acc.items():
        print(f"{mode:7s} {k:62s} launches {len(v):3d}  GRBM_GUI_ACTIVE / ns = {sum(v)/len(v):6.3f}  (divide by the number of XCD counters summed for GHz)")
PY
