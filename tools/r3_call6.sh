#!/bin/bash
# round 3, GPU call 6: in-kernel LayerNorm statistics (no reduction launch), grouped decoder GEMMs on the large-tile kernel
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r3c6 && O=gpurun_out/r3c6
export PYTHONWARNINGS=ignore
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm" 2>&1 | tail -15 > $O/kernels.txt
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_infer_gpu.py -x -q -m gpu -k "bs8 or batch8 or headline or taps or properties" 2>&1 | tail -8 > $O/parity.txt
for i in 1 2; do
  for cfg in "0 0" "1 0" "1 1"; do
    set -- $cfg
    UNIDEPTH_LN_FOLD=$1 UNIDEPTH_GRP_BIG=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['roofline_enc_attention_mlp']; kb=d['kernel_breakdown']
print('fold=$1 grp=$2', d['value'], d['ms_per_step'], 'p50', d['p50_latency_ms'], 'enc scope', e['ms_per_step'], e['frac'], 'launches', e.get('launches_per_step'),
  {k: kb[k]['ms_per_step'] for k in kb if k.startswith('enc.')}, 'dec', round(sum(v['ms_per_step'] for k,v in kb.items() if not k.startswith('enc.'))-e['ms_per_step'],3))" >> $O/ab.txt 2>&1
  done
done
tail -8 $O/kernels.txt; tail -6 $O/parity.txt; cat $O/ab.txt
