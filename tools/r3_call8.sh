#!/bin/bash
# round 3, GPU call 8: 8-wave attention workgroups (A/B against the previous build), max_stack in the fc2 epilogue (V1), grouped fp16 launches
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r3c8 && O=gpurun_out/r3c8
export PYTHONWARNINGS=ignore
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attention or grouped or layernorm_fold" 2>&1 | tail -6 > $O/kernels.txt
timeout 900 python -m pytest tests/test_v1_gpu.py -x -q -m gpu -k "convnext or infer_vs_oracle" 2>&1 | tail -6 > $O/v1.txt
for i in 1 2; do
  UNIDEPTH_HIP_LIB=$PWD/ab/libprev.so UNIDEPTH_HIP_LIB_ALLOW_OLDER=1 timeout 200 python tools/bench_attn.py 2>&1 | tail -2 | sed 's/^/prev /' >> $O/attn.txt
  timeout 200 python tools/bench_attn.py 2>&1 | tail -2 | sed 's/^/new  /' >> $O/attn.txt
done
for i in 1 2; do
  for lib in ab/libprev.so unidepth_amd/libunidepth_hip.so; do
    UNIDEPTH_HIP_LIB=$PWD/$lib UNIDEPTH_HIP_LIB_ALLOW_OLDER=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['roofline_enc_attention_mlp']; kb=d['kernel_breakdown']
print('$lib', d['value'], d['ms_per_step'], 'p50', d['p50_latency_ms'], 'enc scope', e['ms_per_step'], e['frac'], {k: kb[k]['ms_per_step'] for k in kb if k.startswith('enc.')})" >> $O/ab.txt 2>&1
  done
done
timeout 300 python tools/bench_v1.py 16 --no-cpu > $O/v1_bench.txt 2>&1
tail -4 $O/kernels.txt; tail -4 $O/v1.txt; cat $O/attn.txt $O/ab.txt; head -c 500 $O/v1_bench.txt
