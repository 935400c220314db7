#!/bin/bash
# round 3, GPU call 14: ConvNeXt fc1 single fp16 (default) against every weight split (UNIDEPTH_V1_WSPLIT=all): parity numbers + timing A/B
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && exec > gpurun_out/call14.log 2>&1
echo "=== parity, default placement"; timeout 900 python -m pytest tests/test_v1_gpu.py -x -q -s -k "infer_vs_oracle or config4 or convnext_encoder" 2>&1 | grep -v "^$" | tail -25
echo "=== timing A/B (interleaved)"
for r in 1 2; do
  for m in all 1; do echo "WSPLIT=$m"; UNIDEPTH_V1_WSPLIT=$m timeout 300 python tools/bench_v1.py 16 --no-cpu 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo; done
done
