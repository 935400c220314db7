#!/bin/bash
# round 4, GPU call 2: kernel tests (3-deep weight ring), A/B of the weight ring x buffer aliasing on the bench line, in-situ table again
cd "$(dirname "$0")/../.." && R=$PWD && O=gpurun_out/r4c2 && mkdir -p $O
export PYTHONWARNINGS=ignore
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -8 > $O/tests_kernels.txt
timeout 600 python -m pytest tests/test_infer_gpu.py -x -q -m gpu 2>&1 | tail -8 > $O/tests_infer.txt
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['p50_latency_ms'])
except Exception as e: print('$1 FAILED', e)"; }
for r in 1 2; do
  for cfg in "0 0" "1 0" "0 1" "1 1"; do
    set -- $cfg
    UD_GEMM_W3=$1 UNIDEPTH_ALIAS=$2 timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>$O/err.txt | line "w3=$1 alias=$2" >> $O/ab.txt
  done
done
timeout 400 python tools/r4_insitu.py > $O/insitu.txt 2>&1
UD_GEMM_W3=0 UNIDEPTH_ALIAS=0 timeout 400 python tools/r4_insitu.py > $O/insitu_base.txt 2>&1
cat $O/tests_kernels.txt $O/tests_infer.txt $O/ab.txt; grep -v JSON $O/insitu.txt; echo BASE; grep -v JSON $O/insitu_base.txt
