#!/bin/bash
# A/B two library builds inside one GPU-box session (box-to-box variance is larger than most kernel-level deltas)
for i in 1 2; do
for lib in ab/libold.so unidepth_amd/libunidepth_hip.so; do
  UNIDEPTH_HIP_LIB=$PWD/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'])"
done; done
