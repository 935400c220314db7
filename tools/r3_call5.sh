#!/bin/bash
# round 3, GPU call 5: the whole GPU suite, rocprofv3 profiles of the r03 build (kernel stats + FETCH / WRITE passes), per-op dump
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r3c5 && O=gpurun_out/r3c5
export PYTHONWARNINGS=ignore
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $O/pytest_gpu.txt
timeout 300 python tools/bench_ln_fold.py > $O/ln_fold_iso.txt 2>&1
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --dump-ops $O/ops.tsv > $O/bench.txt 2>&1
bash tools/profile_bench.sh r03 > $O/profile.log 2>&1
cat $O/pytest_gpu.txt; cat $O/ln_fold_iso.txt; head -c 700 $O/bench.txt; echo; tail -5 $O/profile.log
