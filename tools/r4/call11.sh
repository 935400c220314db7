#!/bin/bash
# round 4, GPU call 11: operand-power probe (attention + the encoder GEMMs on random / constant / zero operands), stream priority and
# three calls in flight on the bench line, interleaved
cd "$(dirname "$0")/../.." && R=$PWD && O=gpurun_out/r4c11 && mkdir -p $O
export PYTHONWARNINGS=ignore
t0=$(date +%s)
timeout 300 python tools/r4_attn_power.py 2>&1 | grep -v amdgpu.ids > $O/power.txt
echo "[power probe done $(( $(date +%s) - t0 )) s]" >> $O/power.txt
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['p50_latency_ms'])
except Exception as e: print('$1 FAILED', e)"; }
B="python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing"
for r in 1 2; do
  timeout 300 $B 2>$O/err.txt | line "base" >> $O/bench_ab.txt
  UNIDEPTH_PIPE_PRIO=1 timeout 300 $B 2>$O/err.txt | line "prio" >> $O/bench_ab.txt
  timeout 300 $B --inflight 3 2>$O/err.txt | line "inflight3" >> $O/bench_ab.txt
done
echo "[bench ab done $(( $(date +%s) - t0 )) s]" >> $O/bench_ab.txt
cat $O/power.txt $O/bench_ab.txt
