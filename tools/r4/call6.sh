#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD && O=gpurun_out/r4c6 && mkdir -p $O
export PYTHONWARNINGS=ignore
timeout 1700 python -m pytest tests/test_parity_sweep_gpu.py -q -s -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" > $O/sweep.txt
tail -150 $O/sweep.txt
