#!/bin/bash
# round 4, GPU call 9: attention with TWO query blocks per wave (UD_ATTN_QB=2: every K / V^T fragment read feeds two MFMAs, 256 query rows
# per staged tile, 2 waves per SIMD) against the product kernel -- isolated A/B with the correctness check, the kernel tests on the variant
# library, and the bench line with either library, interleaved
cd "$(dirname "$0")/../.." && R=$PWD && O=gpurun_out/r4c9 && mkdir -p $O
export PYTHONWARNINGS=ignore
t0=$(date +%s)
timeout 200 python -m pytest tests/test_v1_gpu.py -q -m gpu -k "three_term" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -4 > $O/fixed_tests.txt
timeout 500 python tools/r4_attn_ab.py --rounds 2 product qb2 qb2_noprio qb2_st3 2>&1 | grep -v amdgpu.ids > $O/attn_ab.txt
echo "[ab done $(( $(date +%s) - t0 )) s]" >> $O/attn_ab.txt
UNIDEPTH_HIP_LIB=$R/ab/libattn_qb2.so timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -6 > $O/attn_tests_qb2.txt
echo "[attention tests on the variant done $(( $(date +%s) - t0 )) s]" >> $O/attn_tests_qb2.txt
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['p50_latency_ms'])
except Exception as e: print('$1 FAILED', e)"; }
for r in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>$O/err.txt | line "product" >> $O/bench_ab.txt
  UNIDEPTH_HIP_LIB=$R/ab/libattn_qb2.so timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>$O/err.txt | line "qb2" >> $O/bench_ab.txt
done
echo "[bench ab done $(( $(date +%s) - t0 )) s]" >> $O/bench_ab.txt
cat $O/fixed_tests.txt $O/attn_ab.txt $O/attn_tests_qb2.txt $O/bench_ab.txt
