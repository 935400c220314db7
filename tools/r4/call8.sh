#!/bin/bash
# round 4, GPU call 8: the two fixed tests, where a tile's time goes inside the attention kernel (stamped build), V1 sweeps (ViT-L/14 with the
# three-term tail; ConvNeXt-L with the fc1 weights split as well), the V2 sweep (timing with the 32-thread oracle + shared encoder pass)
cd "$(dirname "$0")/../.." && R=$PWD && O=gpurun_out/r4c8 && mkdir -p $O
export PYTHONWARNINGS=ignore
t0=$(date +%s)
timeout 300 python -m pytest tests/test_infer_gpu.py tests/test_v1_gpu.py -q -s -m gpu -k "interrupted or three_term" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -12 > $O/fixed_tests.txt
echo "[fixed tests done $(( $(date +%s) - t0 )) s]" >> $O/fixed_tests.txt
UNIDEPTH_HIP_LIB=$R/ab/libattn_trace.so timeout 200 python tools/r4_attn_trace.py 2>&1 | grep -v amdgpu.ids > $O/attn_trace.txt
echo "[trace done $(( $(date +%s) - t0 )) s]" >> $O/attn_trace.txt
timeout 400 python -m pytest tests/test_parity_sweep_gpu.py -q -s -m gpu -k "v1 and vitl14" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -40 > $O/sweep_v1_vitl.txt
echo "[v1 vitl sweep done $(( $(date +%s) - t0 )) s]" >> $O/sweep_v1_vitl.txt
UNIDEPTH_V1_WSPLIT=all timeout 300 python -m pytest tests/test_parity_sweep_gpu.py -q -s -m gpu -k "v1 and cnvnxtl" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -30 > $O/sweep_v1_cnv_wsplit_all.txt
echo "[v1 cnv all-split sweep done $(( $(date +%s) - t0 )) s]" >> $O/sweep_v1_cnv_wsplit_all.txt
timeout 600 python -m pytest tests/test_parity_sweep_gpu.py -q -s -m gpu -k "v2" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -70 > $O/sweep_v2.txt
echo "[v2 sweep done $(( $(date +%s) - t0 )) s]" >> $O/sweep_v2.txt
cat $O/fixed_tests.txt $O/attn_trace.txt; tail -32 $O/sweep_v1_vitl.txt; tail -28 $O/sweep_v1_cnv_wsplit_all.txt; tail -60 $O/sweep_v2.txt
