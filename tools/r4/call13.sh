#!/bin/bash
# round 4, GPU call 13: head convolution with its weights in registers (conv_head_regw_kernel) against the LDS-streamed form
cd "$(dirname "$0")/../.." && R=$PWD && O=gpurun_out/r4c13 && mkdir -p $O
export PYTHONWARNINGS=ignore
t0=$(date +%s)
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "head" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -8 > $O/tests.txt
timeout 400 python -m pytest tests/test_infer_gpu.py -q -m gpu -k "golden or headline" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -8 >> $O/tests.txt
# bit identity of the whole model between the two forms (same MFMA order per accumulator)
timeout 200 python - > $O/bits.txt 2>&1 <<'P'
import os, subprocess, sys, hashlib
code = """
import torch, hashlib, sys
sys.path.insert(0, '.')
from oracle import synth
from unidepth_amd import UniDepthV2
cfg = synth.load_config('vitl14'); sd = synth.make_synthetic_checkpoint(cfg, 125)
m = UniDepthV2(cfg).load_state_dict(sd).to('cuda').eval()
rgb = torch.randint(0, 256, (2, 3, 518, 518), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).cuda()
o = m.infer(rgb); torch.cuda.synchronize()
print(hashlib.sha1(o['depth'].cpu().numpy().tobytes()).hexdigest(), hashlib.sha1(o['confidence'].cpu().numpy().tobytes()).hexdigest())
"""
for v in ("0", "1"):
    e = dict(os.environ, UD_HEAD_REGW=v)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
    print("UD_HEAD_REGW=" + v, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:])
P
echo "[tests done $(( $(date +%s) - t0 )) s]" >> $O/tests.txt
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['p50_latency_ms'])
except Exception as e: print('$1 FAILED', e)"; }
B="python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing"
for r in 1 2 3; do
  UD_HEAD_REGW=0 timeout 300 $B 2>$O/err.txt | line "lds_streamed" >> $O/bench_ab.txt
  timeout 300 $B 2>$O/err.txt | line "register_weights" >> $O/bench_ab.txt
done
echo "[bench ab done $(( $(date +%s) - t0 )) s]" >> $O/bench_ab.txt
cat $O/tests.txt $O/bits.txt $O/bench_ab.txt
