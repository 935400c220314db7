#!/bin/bash
# The batched GPU-box sessions of round 6, one function per gpurun call; every profiles/r06_* file names the session that produced it.
# usage (on the GPU box, through gpurun):  bash tools/r6/sessions.sh <name>
cd "$(dirname "$0")/../.." && R=$PWD
export PYTHONWARNINGS=ignore
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('p50_latency_ms'), d.get('value_one_call'))
except Exception as e: print('$1 FAILED', e)"; }

# round 6, GPU call 1: the product's K loop on plain 4096^3 / 8192^3 / encoder-shape problems against the ping-pong
# (staggered 8-phase, s_setprio) form of the same tile (tools/ubench/gemm8p.hip) on the same operand fill and box; the pure-MFMA stream
# of tools/ubench/gemm4w for the box's attainable rate; a bench line for the box's level
call1() {
O=gpurun_out/r6c1 && mkdir -p $O
t0=$(date +%s)
for s in "4096 4096 4096" "8192 8192 8192" "11008 1024 1024" "11008 3072 1024" "11008 4096 1024" "11008 1024 4096"; do
  timeout 120 tools/ubench/gemm8p $s 2>&1 | grep -v amdgpu.ids >> $O/gemm8p.txt
done
echo "[gemm8p done $(( $(date +%s) - t0 )) s]" >> $O/gemm8p.txt
timeout 300 python tools/bench_gemm_plain.py 2>&1 | grep -v amdgpu.ids > $O/product_plain.txt
echo "[product done $(( $(date +%s) - t0 )) s]" >> $O/product_plain.txt
timeout 120 tools/ubench/gemm4w 16384 4096 4096 2>&1 | grep -v amdgpu.ids > $O/gemm4w.txt
timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs 2>$O/err.txt | tee $O/bench.json | line "bench" > $O/bench.txt
cat $O/gemm8p.txt $O/product_plain.txt $O/gemm4w.txt $O/bench.txt; tail -3 $O/err.txt
}

# round 6, GPU call 2: the same A/B in ONE process (the product's kernel through its C-ABI from the micro-benchmark, arms interleaved: call 1
# showed the same product kernel 9 % apart between two places of one python loop -- clock ramp); --inflight 1 / 2 / 3 / 4 on the bench line
call2() {
O=gpurun_out/r6c2 && mkdir -p $O
for s in "4096 4096 4096" "11008 1024 1024" "11008 1024 4096" "11008 3072 1024" "11008 4096 1024"; do
  UD_LIB=$R/unidepth_amd/libunidepth_hip.so timeout 120 tools/ubench/gemm8p $s 2>&1 | grep -v amdgpu.ids >> $O/gemm8p_vs_product.txt
done
for r in 1 2; do for inf in 2 3 4; do
  timeout 300 python bench.py --steps 24 --warmup 4 --inflight $inf --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>$O/err.txt | line "inflight=$inf" >> $O/inflight.txt
done; done
cat $O/gemm8p_vs_product.txt $O/inflight.txt; tail -3 $O/err.txt
}

"$@"
