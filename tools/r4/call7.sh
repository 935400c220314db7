#!/bin/bash
# round 4, GPU call 7: new tests (hipGraph replay, interrupted replay, three-term V1 tail), the V1 ConvNeXt sweep with / without the third
# term, the bench line with its new sub-records (latency_bs1) and an A/B of graph replay on the headline
cd "$(dirname "$0")/../.." && R=$PWD && O=gpurun_out/r4c7 && mkdir -p $O
export PYTHONWARNINGS=ignore
t0=$(date +%s)
timeout 400 python -m pytest tests/test_infer_gpu.py -q -s -m gpu -k "interrupted or graph_replay" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -25 > $O/new_tests.txt
echo "[new tests done $(( $(date +%s) - t0 )) s]" >> $O/new_tests.txt
timeout 600 python -m pytest tests/test_v1_gpu.py -q -s -m gpu -x 2>&1 | grep -v "^$\|amdgpu.ids" | tail -40 > $O/v1_tests.txt
echo "[v1 tests done $(( $(date +%s) - t0 )) s]" >> $O/v1_tests.txt
timeout 500 python -m pytest tests/test_parity_sweep_gpu.py -q -s -m gpu -k "v1 and cnvnxtl" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -60 > $O/sweep_v1_asplit.txt
echo "[sweep done $(( $(date +%s) - t0 )) s]" >> $O/sweep_v1_asplit.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "[bench done $(( $(date +%s) - t0 )) s]" >> $O/bench.err
UNIDEPTH_GRAPH=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-kernel-timing > $O/bench_graph.json 2>> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-kernel-timing > $O/bench_eager2.json 2>> $O/bench.err
echo "[all done $(( $(date +%s) - t0 )) s]" >> $O/bench.err
cat $O/new_tests.txt; tail -30 $O/v1_tests.txt; tail -45 $O/sweep_v1_asplit.txt
python - <<'P'
import json
for f in ("bench", "bench_graph", "bench_eager2"):
    try:
        d = json.loads(open(f"gpurun_out/r4c7/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("p50_latency_ms"), d.get("value_one_call"))
        if "configs" in d:
            print(json.dumps(d["configs"].get("latency_bs1")))
            v1 = d["configs"].get("v1_cnvnxtl_640x480_bs16", {})
            print("v1", v1.get("value"), v1.get("ms_per_step"), v1.get("error"))
        if "roofline" in d: print(json.dumps(d["roofline"]))
    except Exception as e:
        print(f, "unreadable", e)
P
tail -5 $O/bench.err
