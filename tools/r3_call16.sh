#!/bin/bash
# round 3, GPU call 16: the default bench line of the final build (CPU baselines, sub-records), 3 calls in flight against 2, rocprofv3 passes
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r3c16 && O=gpurun_out/r3c16
export PYTHONWARNINGS=ignore
timeout 1200 python bench.py --dump-ops $O/ops.tsv > $O/bench.txt 2> $O/bench.err
for n in 2 3 2 3; do timeout 300 python bench.py --inflight $n --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>/dev/null | grep -o '"value": [0-9.]*, \|"inflight": [0-9]*' | tr '\n' ' '; echo; done > $O/inflight.txt
bash tools/profile_bench.sh r03 > $O/profile.log 2>&1
head -c 1200 $O/bench.txt; echo; cat $O/inflight.txt; tail -3 $O/profile.log
