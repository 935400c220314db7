#!/bin/bash
cd "$(dirname "$0")/../.." && R=$PWD && O=gpurun_out/r4c5 && mkdir -p $O
export PYTHONWARNINGS=ignore
timeout 400 python tools/r4_sweep_diag.py 301 518 518 b8 2>&1 | grep -v "amdgpu.ids" | tail -14 > $O/diag_301_518.txt
timeout 400 python tools/r4_sweep_diag.py 335 518 518 b8 2>&1 | grep -v "amdgpu.ids" | tail -14 > $O/diag_335_518.txt
cat $O/diag_301_518.txt; echo; cat $O/diag_335_518.txt
