#!/bin/bash
# round 3, GPU call 12: folded-LayerNorm consumer with per-tile statistics tables (any number of tiles per workgroup): bs = 8 A/B against the
# previous build, bs = 16 / 32 and the 644x966 shape with the fold on / off
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r3c12 && O=gpurun_out/r3c12
export PYTHONWARNINGS=ignore
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layernorm_fold" 2>&1 | tail -4 > $O/kernels.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -s -k "config5 or 644x966 or headline" 2>&1 | grep -v Warn | tail -8 > $O/parity.txt
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['roofline_enc_attention_mlp']
print('$1', d['value'], d['ms_per_step'], 'p50', d['p50_latency_ms'], 'enc scope', e['ms_per_step'], e['frac'], 'launches', e.get('launches_per_step'))"; }
for i in 1 2; do
  for lib in ab/libprev.so unidepth_amd/libunidepth_hip.so; do
    UNIDEPTH_HIP_LIB=$PWD/$lib UNIDEPTH_HIP_LIB_ALLOW_OLDER=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs 2>/dev/null | line "bs8 $lib" >> $O/ab.txt 2>&1
  done
done
for bs in 16 32; do for f in 0 1; do
  UNIDEPTH_LN_FOLD=$f timeout 300 python bench.py --batch $bs --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs 2>/dev/null | line "bs$bs fold=$f" >> $O/ab.txt 2>&1
done; done
for f in 0 1; do
  UNIDEPTH_LN_FOLD=$f timeout 300 python bench.py --batch 4 --size 644 966 --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs 2>/dev/null | line "644x966 bs4 fold=$f" >> $O/ab.txt 2>&1
done
tail -3 $O/kernels.txt; tail -6 $O/parity.txt; cat $O/ab.txt
