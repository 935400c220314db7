#!/usr/bin/env python
"""Slice a rocprofv3 run of tools/r4_insitu.py by phase: per (phase, kernel family) the mean duration and the mean counter value per
ns of kernel time (GRBM_GUI_ACTIVE / ns ~ shader clock in GHz x the number of XCD counter instances summed; FETCH_SIZE in KB x 2 per the
guide's gfx950 correction is left to the reader -- ratios between phases are what this prints).
usage: r4_insitu_post.py <rocprof output dir> <phases.json>"""
import collections, csv, glob, json, sys

d, pj = sys.argv[1], sys.argv[2]
ph = json.load(open(pj))
tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not tr:
    sys.exit("no kernel trace in " + d)
rows = [r for r in csv.DictReader(open(tr[0])) if "anonymous namespace" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
need = sum(n for _, n in ph["phases"])
if len(rows) < need:
    sys.exit(f"trace has {len(rows)} engine launches, phases need {need}")
rows = rows[-need:]
ctr = collections.defaultdict(dict)
if cc:
    for r in csv.DictReader(open(cc[0])):
        ctr[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])


def fam(name):
    for k in ("attention_kernel", "layernorm_kernel"):
        if k in name:
            return k
    i = name.find("gemm")
    return name[i:name.find("(", i)] if i >= 0 else name[:40]


pos = 0
out = []
for pname, n in ph["phases"]:
    agg = collections.defaultdict(lambda: [0, 0.0, collections.defaultdict(float)])
    for r in rows[pos:pos + n]:
        a = agg[fam(r["Kernel_Name"])]
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a[0] += 1; a[1] += dur
        for cn, cv in ctr.get(r["Dispatch_Id"], {}).items():
            a[2][cn] += cv
    pos += n
    for f, (cnt, dur, cs) in agg.items():
        if f == "layernorm_kernel" and not pname.startswith("seq"):
            continue
        line = f"{pname:18s} {f:50s} n={cnt:4d}  avg {dur / cnt / 1e3:7.1f} us"
        for cn, cv in cs.items():
            line += f"  {cn}/ns {cv / dur:8.3f}  {cn}/launch {cv / cnt:12.0f}"
        out.append(line)
print("\n".join(out))
