#!/bin/bash
# round 3, GPU call 11: head conv with all 9 weight slabs staged at once (A/B against the previous build)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r3c11 && O=gpurun_out/r3c11
export PYTHONWARNINGS=ignore
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "head_conv or conv3x3" 2>&1 | tail -4 > $O/kernels.txt
for i in 1 2; do
  for lib in ab/libprev.so unidepth_amd/libunidepth_hip.so; do
    UNIDEPTH_HIP_LIB=$PWD/$lib UNIDEPTH_HIP_LIB_ALLOW_OLDER=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs 2>$O/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['roofline_enc_attention_mlp']; kb=d['kernel_breakdown']
print('$lib', d['value'], d['ms_per_step'], 'p50', d['p50_latency_ms'], 'enc scope', e['ms_per_step'], 'hr conv', kb.get('conv_tile_kernel<2, 4, true, true>'))" >> $O/ab.txt 2>&1
  done
done
tail -3 $O/kernels.txt; cat $O/ab.txt; tail -3 $O/err.txt
