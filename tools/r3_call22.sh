#!/bin/bash
# round 3, GPU call 22: attention with the tile's softmax split in two key-block halves, the second interleaved with the first half's P V MFMAs
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && exec > gpurun_out/call22.log 2>&1
export PYTHONWARNINGS=ignore
echo "=== attention tests, split-softmax build"; UNIDEPTH_HIP_LIB=$PWD/ab/libsm.so timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -3
echo "=== bench_attn (interleaved)"
for r in 1 2 3; do for lib in unidepth_amd/libunidepth_hip.so ab/libsm.so; do echo -n "$lib  "; UNIDEPTH_HIP_LIB=$PWD/$lib timeout 120 python tools/bench_attn.py 2>&1 | tail -1; done; done
echo "=== bench.py (interleaved)"
for r in 1 2; do for lib in unidepth_amd/libunidepth_hip.so ab/libsm.so; do echo -n "$lib  "
  UNIDEPTH_HIP_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>/dev/null | grep -o '"value": [0-9.]*, \|"p50_latency_ms": [0-9.]*' | tr '\n' ' '; echo; done; done
