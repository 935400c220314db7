#!/bin/bash
# round 3, GPU call 17: Nystrom stages as flash attention (split-key kernel_3 v, landmark-key output stage): kernel tests, V1 parity, timing A/B
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && exec > gpurun_out/call17.log 2>&1
export PYTHONWARNINGS=ignore
echo "=== kernel tests"; timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or transpose_to_attention" 2>&1 | tail -6
echo "=== V1 parity"; timeout 900 python -m pytest tests/test_v1_gpu.py -x -q -s -k "infer_vs_oracle or decoder_taps" 2>&1 | grep -v "^$" | grep -v Warning | tail -14
echo "=== timing A/B (interleaved)"
for r in 1 2; do
  for m in 0 1; do echo "NYS_FLASH=$m"; UNIDEPTH_V1_NYS_FLASH=$m timeout 300 python tools/bench_v1.py 16 --no-cpu 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"launches": [0-9]*\|Error.*' | tr '\n' ' '; echo; done
done
UNIDEPTH_V1_NYS_FLASH=1 timeout 300 python tools/bench_v1.py 16 --no-cpu --dump 2>&1 | grep -n "nys\|softmax\|landmarks" | head -60
