#!/bin/bash
# round 4, GPU call 16: head conv kernel with the padded halo pitch + one-group software pipeline of the fragment reads, against its first form
cd "$(dirname "$0")/../.." && R=$PWD && O=gpurun_out/r4c16 && mkdir -p $O
export PYTHONWARNINGS=ignore
t0=$(date +%s)
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "head" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -6 > $O/tests.txt
timeout 500 python -m pytest tests/test_infer_gpu.py tests/test_parity_gpu.py -q -m gpu -k "golden or headline or taps" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -6 >> $O/tests.txt
echo "[tests done $(( $(date +%s) - t0 )) s]" >> $O/tests.txt
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['p50_latency_ms'])
except Exception as e: print('$1 FAILED', e)"; }
B="python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing"
for r in 1 2 3; do
  UNIDEPTH_HIP_LIB=$R/ab/libhead_prev.so timeout 300 $B 2>$O/err.txt | line "first_form" >> $O/bench_ab.txt
  timeout 300 $B 2>$O/err.txt | line "padded_pipelined" >> $O/bench_ab.txt
done
echo "[bench ab done $(( $(date +%s) - t0 )) s]" >> $O/bench_ab.txt
cat $O/tests.txt $O/bench_ab.txt
