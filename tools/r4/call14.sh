#!/bin/bash
# round 4, GPU call 14: the end-of-round bench line again with the kernel-class labels that match this build's rocprofv3 names (traffic
# from profiles/r04_hbm_traffic.json), per-launch table
cd "$(dirname "$0")/../.." && R=$PWD && O=gpurun_out/r4c14 && mkdir -p $O
export PYTHONWARNINGS=ignore
timeout 900 python bench.py --dump-ops $O/ops_per_launch.tsv > $O/bench.json 2> $O/bench.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/r4c14/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "p50_latency_ms", "value_one_call")})
print(json.dumps(d["roofline"])[:700])
print(json.dumps(d.get("roofline_enc_attention_mlp"))[:500])
for k, v in d.get("configs", {}).items():
    print(k, v.get("value"), v.get("ms_per_step"), v.get("error"), json.dumps(v)[:300] if k == "latency_bs1" else "")
print(json.dumps(d.get("kernel_breakdown", {}))[:1500])
P
