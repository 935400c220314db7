#!/bin/bash
# round 3, GPU call 19: hi / lo K-tiles interleaved in the two-term products: kernel tests, V1 parity, A/B against the previous gemm.o (V1 and V2)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && exec > gpurun_out/call19.log 2>&1
export PYTHONWARNINGS=ignore
echo "=== kernel tests (gemm)"; timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or split or wrap or conv" 2>&1 | tail -4
echo "=== V1 parity"; timeout 900 python -m pytest tests/test_v1_gpu.py -x -q -s -k "infer_vs_oracle or convnext_encoder" 2>&1 | grep -v "^$" | grep -v Warning | tail -10
echo "=== A/B V1"
for r in 1 2; do for lib in ab/libhead.so unidepth_amd/libunidepth_hip.so; do echo -n "$lib  "
  UNIDEPTH_HIP_LIB=$PWD/$lib timeout 300 python tools/bench_v1.py 16 --no-cpu 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"encoder_ms": [0-9.]*\|"decoder_ms": [0-9.]*\|Error.*' | tr '\n' ' '; echo; done; done
echo "=== A/B V2"
for r in 1 2; do for lib in ab/libhead.so unidepth_amd/libunidepth_hip.so; do echo -n "$lib  "
  UNIDEPTH_HIP_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>/dev/null | grep -o '"value": [0-9.]*, \|"p50_latency_ms": [0-9.]*' | tr '\n' ' '; echo; done; done
