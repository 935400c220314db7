#!/bin/bash
# round 3, GPU call 18: SH embedding (token per lane), landmark pooling (token slices), pinv chain with the dual-output product: tests + timing
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && exec > gpurun_out/call18.log 2>&1
export PYTHONWARNINGS=ignore
echo "=== V1 kernel + parity tests"; timeout 1200 python -m pytest tests/test_v1_gpu.py -x -q -s -k "not config4 and not convnext_encoder" 2>&1 | grep -v "^$" | grep -v Warning | tail -16
echo "=== timing"
for r in 1 2; do timeout 300 python tools/bench_v1.py 16 --no-cpu 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"launches": [0-9]*\|Error.*' | tr '\n' ' '; echo; done
timeout 300 python tools/bench_v1.py 16 --no-cpu --dump 2>&1 | grep "sh_embed\|landmarks\|pinv  " | head -12
