#!/bin/bash
# round 3, GPU call 7: the default bench line (CPU baselines, sub-records of the other configs) + rocprofv3 passes of the final build
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r3c7 && O=gpurun_out/r3c7
export PYTHONWARNINGS=ignore
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_v1_gpu.py -x -q -m gpu -k "layernorm_fold or grouped or v1_infer" 2>&1 | tail -6 > $O/tests.txt
timeout 1200 python bench.py --dump-ops $O/ops.tsv > $O/bench.txt 2> $O/bench.err
bash tools/profile_bench.sh r03 > $O/profile.log 2>&1
cat $O/tests.txt; head -c 1500 $O/bench.txt; echo; tail -3 $O/profile.log
