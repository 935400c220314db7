#!/bin/bash
# round 4, GPU call 4: diagnose the checkpoint seed that misses the bar (taps), with the round-4 switches on and off
cd "$(dirname "$0")/../.." && R=$PWD && O=gpurun_out/r4c4 && mkdir -p $O
export PYTHONWARNINGS=ignore
timeout 300 python tools/r4_sweep_diag.py 301 644 966 > $O/diag_301_644.txt 2>&1
UD_GEMM_W3=0 UNIDEPTH_ALIAS=0 timeout 300 python tools/r4_sweep_diag.py 301 644 966 2>&1 | head -3 > $O/diag_301_644_base.txt
timeout 300 python tools/r4_sweep_diag.py 301 518 518 > $O/diag_301_518.txt 2>&1
timeout 300 python tools/r4_sweep_diag.py 318 518 518 > $O/diag_318_518.txt 2>&1
cat $O/diag_301_644.txt; echo BASE; cat $O/diag_301_644_base.txt; echo; cat $O/diag_301_518.txt; echo; cat $O/diag_318_518.txt
