#!/bin/bash
# round 4, GPU call 12: the decoder's camera branch on the program's side stream -- tests (bit identity eager / graph / taps / pipeline),
# the tap-level parity tests, and the bench line with and without it, interleaved (the one-call p50 is the number it is for)
cd "$(dirname "$0")/../.." && R=$PWD && O=gpurun_out/r4c12 && mkdir -p $O
export PYTHONWARNINGS=ignore
t0=$(date +%s)
timeout 400 python -m pytest tests/test_infer_gpu.py -q -m gpu -k "side_branch or graph_replay or interrupted or pipeline" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -12 > $O/tests.txt
timeout 400 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "taps or seams or camera_batch or warm_state" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -8 >> $O/tests.txt
echo "[tests done $(( $(date +%s) - t0 )) s]" >> $O/tests.txt
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['p50_latency_ms'], d.get('p90_latency_ms'))
except Exception as e: print('$1 FAILED', e)"; }
B="python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing"
for r in 1 2 3; do
  UNIDEPTH_SIDE=0 timeout 300 $B 2>$O/err.txt | line "one_stream" >> $O/bench_ab.txt
  timeout 300 $B 2>$O/err.txt | line "side_branch" >> $O/bench_ab.txt
done
echo "[bench ab done $(( $(date +%s) - t0 )) s]" >> $O/bench_ab.txt
cat $O/tests.txt $O/bench_ab.txt
