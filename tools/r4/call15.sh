#!/bin/bash
# round 4, GPU call 15: x2 up-sampling fused into the ConvTranspose accumulate (UdGemm.up_src) against the materialised form
cd "$(dirname "$0")/../.." && R=$PWD && O=gpurun_out/r4c15 && mkdir -p $O
export PYTHONWARNINGS=ignore
t0=$(date +%s)
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "d2s" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -8 > $O/tests.txt
timeout 500 python -m pytest tests/test_parity_gpu.py tests/test_infer_gpu.py -q -m gpu -k "taps or golden or headline or more_shapes or seams" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -8 >> $O/tests.txt
echo "[tests done $(( $(date +%s) - t0 )) s]" >> $O/tests.txt
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['p50_latency_ms'])
except Exception as e: print('$1 FAILED', e)"; }
B="python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing"
for r in 1 2 3; do
  UNIDEPTH_UPFUSE=0 timeout 300 $B 2>$O/err.txt | line "materialised" >> $O/bench_ab.txt
  timeout 300 $B 2>$O/err.txt | line "fused" >> $O/bench_ab.txt
done
echo "[bench ab done $(( $(date +%s) - t0 )) s]" >> $O/bench_ab.txt
cat $O/tests.txt $O/bench_ab.txt
