#!/bin/bash
# round 3, GPU call 21: UniDepthV1 at other operating points (ViT-L/14 backbone; batch 1 / 4 latency) for the README table
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && exec > gpurun_out/call21.log 2>&1
export PYTHONWARNINGS=ignore
for a in "16 --vitl14" "1" "4" "1 --vitl14"; do echo "bench_v1 $a"; timeout 300 python tools/bench_v1.py $a --no-cpu 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"launches": [0-9]*\|"encoder_ms": [0-9.]*\|"decoder_ms": [0-9.]*\|Error.*' | tr '\n' ' '; echo; done
