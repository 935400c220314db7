#!/bin/bash
# round 4, GPU call 3: attention segment trace; parity sweep (8 seeds x 2 sizes x 3 model families)
cd "$(dirname "$0")/../.." && R=$PWD && O=gpurun_out/r4c3 && mkdir -p $O
export PYTHONWARNINGS=ignore
UNIDEPTH_HIP_LIB=$R/ab/libattn_trace.so timeout 200 python tools/r4_attn_trace.py > $O/attn_trace.txt 2>&1
timeout 1500 python -m pytest tests/test_parity_sweep_gpu.py -x -q -s -m gpu 2>&1 | grep -v "^$" | tail -120 > $O/sweep.txt
cat $O/attn_trace.txt; cat $O/sweep.txt
