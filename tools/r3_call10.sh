#!/bin/bash
# round 3, GPU call 10: statistics + ticket ahead of the row stores in the producers' straight-line epilogue (A/B against the previous build)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r3c10 && O=gpurun_out/r3c10
export PYTHONWARNINGS=ignore
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layernorm_fold or grouped" 2>&1 | tail -5 > $O/kernels.txt
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_infer_gpu.py -x -q -m gpu -s -k "config5 or headline" 2>&1 | grep -v Warn | tail -8 > $O/parity.txt
for i in 1 2; do
  UNIDEPTH_HIP_LIB=$PWD/ab/libprev.so UNIDEPTH_HIP_LIB_ALLOW_OLDER=1 timeout 200 python tools/bench_ln_fold.py 2>&1 | grep "producer\|classic" | grep "proj\|fc2" | tr '\n' ' ' | sed 's/^/prev /' >> $O/prod.txt; echo >> $O/prod.txt
  timeout 200 python tools/bench_ln_fold.py 2>&1 | grep "producer\|classic" | grep "proj\|fc2" | tr '\n' ' ' | sed 's/^/new  /' >> $O/prod.txt; echo >> $O/prod.txt
done
for i in 1 2; do
  for lib in ab/libprev.so unidepth_amd/libunidepth_hip.so; do
    UNIDEPTH_HIP_LIB=$PWD/$lib UNIDEPTH_HIP_LIB_ALLOW_OLDER=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs 2>$O/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['roofline_enc_attention_mlp']; kb=d['kernel_breakdown']
print('$lib', d['value'], d['ms_per_step'], 'p50', d['p50_latency_ms'], 'enc scope', e['ms_per_step'], e['frac'], {k: kb[k]['ms_per_step'] for k in kb if k.startswith('enc.')})" >> $O/ab.txt 2>&1
  done
done
tail -3 $O/kernels.txt; tail -6 $O/parity.txt; cat $O/prod.txt $O/ab.txt; tail -3 $O/err.txt
