#!/bin/bash
# round 3, GPU call 15: is UniDepthV1 slower with the current library than with the one of the profile run (commit d63d2a6)?  interleaved
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && exec > gpurun_out/call15.log 2>&1
for r in 1 2; do
  for lib in ab/libd63.so unidepth_amd/libunidepth_hip.so; do echo "lib=$lib"
    UNIDEPTH_HIP_LIB=$PWD/$lib UNIDEPTH_HIP_LIB_ALLOW_OLDER=1 UNIDEPTH_V1_WSPLIT=all timeout 300 python tools/bench_v1.py 16 --no-cpu 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|Error.*' | tr '\n' ' '; echo; done
done
