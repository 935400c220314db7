#!/bin/bash
# round 3, GPU call 3: LayerNorm fold v5 (finalize kernel + LDS tables): kernel tests, ViT-L parity, interleaved A/B of the step
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r3c4 && O=gpurun_out/r3c4
export PYTHONWARNINGS=ignore
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layernorm_fold or gemm_big or row_balanced" 2>&1 | tail -15 > $O/kernels.txt
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "bs8 or batch8 or headline or vitl" 2>&1 | tail -5 > $O/parity.txt; timeout 300 python tools/bench_ln_fold.py > $O/ln_fold_iso.txt 2>&1
for i in 1 2; do
  for f in 0 1; do
    UNIDEPTH_LN_FOLD=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['roofline_enc_attention_mlp']; kb=d['kernel_breakdown']
print('fold=$f', d['value'], d['ms_per_step'], 'p50', d['p50_latency_ms'], 'p90', d['p90_latency_ms'], 'enc scope', e['ms_per_step'], e['frac'], 'launches', e.get('launches_per_step'),
  {k: kb[k]['ms_per_step'] for k in kb if k.startswith('enc.')})" >> $O/ab_fold.txt 2>&1
  done
done
cat $O/ln_fold_iso.txt; tail -6 $O/kernels.txt; tail -6 $O/parity.txt; cat $O/ab_fold.txt
