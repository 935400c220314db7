#!/bin/bash
# HISTORICAL (round 4): several switches set below (UD_GEMM_W3, UNIDEPTH_ALIAS, UNIDEPTH_SIDE, UNIDEPTH_GRAPH, UD_HEAD_REGW, UNIDEPTH_UPFUSE,
# UNIDEPTH_PIPE_PRIO) and the attention compile-time variants were removed in round 5; kept as the record of how profiles/r04_* were produced.
# The batched GPU-box sessions of round 4, one function per gpurun call (tests + interleaved A/B + profiles in one call each); every
# profiles/r04_* file names the session that produced it.  usage (on the GPU box, through gpurun):  bash tools/r4/sessions.sh <name>
#   e.g.  /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r4/sessions.sh final'
# (function bodies are not indented: they contain here-documents)
cd "$(dirname "$0")/../.." && R=$PWD
export PYTHONWARNINGS=ignore

# round 4, GPU call 1 (diagnosis): MFMA/VALU overlap micro-benchmark, attention variants A/B, isolated vs in-situ encoder launches
# (plain, then under rocprofv3 with GRBM_GUI_ACTIVE and FETCH_SIZE), a short baseline bench line for this box.
call1() {
O=gpurun_out/r4c1 && mkdir -p $O
timeout 120 tools/ubench/overlap > $O/overlap.txt 2>&1
timeout 600 python tools/r4_attn_ab.py --rounds 2 base noprio nomax nomax_noprio st3 st3_nomax w8st4 w8st4_nomax w8st2_nomax oagpr > $O/attn_ab.txt 2>&1
timeout 400 python tools/r4_insitu.py > $O/insitu.txt 2>&1
( cd /tmp && export TMPDIR=/tmp
  timeout 500 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_clk -o c -- python $R/tools/r4_insitu.py --quick --phases-json $R/$O/phases_clk.json > $R/$O/insitu_clk.log 2>&1
  timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch -o f -- python $R/tools/r4_insitu.py --quick --phases-json $R/$O/phases_fetch.json > $R/$O/insitu_fetch.log 2>&1 )
python tools/r4_insitu_post.py $O/pmc_clk $O/phases_clk.json > $O/insitu_clk.txt 2>&1
python tools/r4_insitu_post.py $O/pmc_fetch $O/phases_fetch.json > $O/insitu_fetch.txt 2>&1
rm -rf $O/pmc_clk $O/pmc_fetch            # raw traces are large; the sliced tables are what is kept
timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-configs > $O/bench.txt 2> $O/bench.err
cat $O/overlap.txt; cat $O/attn_ab.txt; cat $O/insitu.txt | grep -v JSON; echo CLK; cat $O/insitu_clk.txt | cut -c1-220; echo FETCH; cat $O/insitu_fetch.txt | cut -c1-220; head -c 700 $O/bench.txt
}

# round 4, GPU call 2: kernel tests (3-deep weight ring), A/B of the weight ring x buffer aliasing on the bench line, in-situ table again
call2() {
O=gpurun_out/r4c2 && mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu 2>&1 | tail -8 > $O/tests_kernels.txt
timeout 600 python -m pytest tests/test_infer_gpu.py -x -q -m gpu 2>&1 | tail -8 > $O/tests_infer.txt
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['p50_latency_ms'])
except Exception as e: print('$1 FAILED', e)"; }
for r in 1 2; do
  for cfg in "0 0" "1 0" "0 1" "1 1"; do
    set -- $cfg
    UD_GEMM_W3=$1 UNIDEPTH_ALIAS=$2 timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>$O/err.txt | line "w3=$1 alias=$2" >> $O/ab.txt
  done
done
timeout 400 python tools/r4_insitu.py > $O/insitu.txt 2>&1
UD_GEMM_W3=0 UNIDEPTH_ALIAS=0 timeout 400 python tools/r4_insitu.py > $O/insitu_base.txt 2>&1
cat $O/tests_kernels.txt $O/tests_infer.txt $O/ab.txt; grep -v JSON $O/insitu.txt; echo BASE; grep -v JSON $O/insitu_base.txt
}

# round 4, GPU call 3: attention segment trace; parity sweep (8 seeds x 2 sizes x 3 model families)
call3() {
O=gpurun_out/r4c3 && mkdir -p $O
UNIDEPTH_HIP_LIB=$R/ab/libattn_trace.so timeout 200 python tools/r4_attn_trace.py > $O/attn_trace.txt 2>&1
timeout 1500 python -m pytest tests/test_parity_sweep_gpu.py -x -q -s -m gpu 2>&1 | grep -v "^$" | tail -120 > $O/sweep.txt
cat $O/attn_trace.txt; cat $O/sweep.txt
}

# round 4, GPU call 4: diagnose the checkpoint seed that misses the bar (taps), with the round-4 switches on and off
call4() {
O=gpurun_out/r4c4 && mkdir -p $O
timeout 300 python tools/r4_sweep_diag.py 301 644 966 > $O/diag_301_644.txt 2>&1
UD_GEMM_W3=0 UNIDEPTH_ALIAS=0 timeout 300 python tools/r4_sweep_diag.py 301 644 966 2>&1 | head -3 > $O/diag_301_644_base.txt
timeout 300 python tools/r4_sweep_diag.py 301 518 518 > $O/diag_301_518.txt 2>&1
timeout 300 python tools/r4_sweep_diag.py 318 518 518 > $O/diag_318_518.txt 2>&1
cat $O/diag_301_644.txt; echo BASE; cat $O/diag_301_644_base.txt; echo; cat $O/diag_301_518.txt; echo; cat $O/diag_318_518.txt
}

call5() {
O=gpurun_out/r4c5 && mkdir -p $O
timeout 400 python tools/r4_sweep_diag.py 301 518 518 b8 2>&1 | grep -v "amdgpu.ids" | tail -14 > $O/diag_301_518.txt
timeout 400 python tools/r4_sweep_diag.py 335 518 518 b8 2>&1 | grep -v "amdgpu.ids" | tail -14 > $O/diag_335_518.txt
cat $O/diag_301_518.txt; echo; cat $O/diag_335_518.txt
}

call6() {
O=gpurun_out/r4c6 && mkdir -p $O
timeout 1700 python -m pytest tests/test_parity_sweep_gpu.py -q -s -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" > $O/sweep.txt
tail -150 $O/sweep.txt
}

# round 4, GPU call 7: new tests (hipGraph replay, interrupted replay, three-term V1 tail), the V1 ConvNeXt sweep with / without the third
# term, the bench line with its new sub-records (latency_bs1) and an A/B of graph replay on the headline
call7() {
O=gpurun_out/r4c7 && mkdir -p $O
t0=$(date +%s)
timeout 400 python -m pytest tests/test_infer_gpu.py -q -s -m gpu -k "interrupted or graph_replay" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -25 > $O/new_tests.txt
echo "[new tests done $(( $(date +%s) - t0 )) s]" >> $O/new_tests.txt
timeout 600 python -m pytest tests/test_v1_gpu.py -q -s -m gpu -x 2>&1 | grep -v "^$\|amdgpu.ids" | tail -40 > $O/v1_tests.txt
echo "[v1 tests done $(( $(date +%s) - t0 )) s]" >> $O/v1_tests.txt
timeout 500 python -m pytest tests/test_parity_sweep_gpu.py -q -s -m gpu -k "v1 and cnvnxtl" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -60 > $O/sweep_v1_asplit.txt
echo "[sweep done $(( $(date +%s) - t0 )) s]" >> $O/sweep_v1_asplit.txt
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "[bench done $(( $(date +%s) - t0 )) s]" >> $O/bench.err
UNIDEPTH_GRAPH=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-kernel-timing > $O/bench_graph.json 2>> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-extra-configs --no-cpu-baseline --no-kernel-timing > $O/bench_eager2.json 2>> $O/bench.err
echo "[all done $(( $(date +%s) - t0 )) s]" >> $O/bench.err
cat $O/new_tests.txt; tail -30 $O/v1_tests.txt; tail -45 $O/sweep_v1_asplit.txt
python - <<'P'
import json
for f in ("bench", "bench_graph", "bench_eager2"):
    try:
        d = json.loads(open(f"gpurun_out/r4c7/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("p50_latency_ms"), d.get("value_one_call"))
        if "configs" in d:
            print(json.dumps(d["configs"].get("latency_bs1")))
            v1 = d["configs"].get("v1_cnvnxtl_640x480_bs16", {})
            print("v1", v1.get("value"), v1.get("ms_per_step"), v1.get("error"))
        if "roofline" in d: print(json.dumps(d["roofline"]))
    except Exception as e:
        print(f, "unreadable", e)
P
tail -5 $O/bench.err
}

# round 4, GPU call 8: the two fixed tests, where a tile's time goes inside the attention kernel (stamped build), V1 sweeps (ViT-L/14 with the
# three-term tail; ConvNeXt-L with the fc1 weights split as well), the V2 sweep (timing with the 32-thread oracle + shared encoder pass)
call8() {
O=gpurun_out/r4c8 && mkdir -p $O
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
}

# round 4, GPU call 9: attention with TWO query blocks per wave (UD_ATTN_QB=2: every K / V^T fragment read feeds two MFMAs, 256 query rows
# per staged tile, 2 waves per SIMD) against the product kernel -- isolated A/B with the correctness check, the kernel tests on the variant
# library, and the bench line with either library, interleaved
call9() {
O=gpurun_out/r4c9 && mkdir -p $O
t0=$(date +%s)
timeout 200 python -m pytest tests/test_v1_gpu.py -q -m gpu -k "three_term" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -4 > $O/fixed_tests.txt
timeout 500 python tools/r4_attn_ab.py --rounds 2 product qb2 qb2_noprio qb2_st3 2>&1 | grep -v amdgpu.ids > $O/attn_ab.txt
echo "[ab done $(( $(date +%s) - t0 )) s]" >> $O/attn_ab.txt
UNIDEPTH_HIP_LIB=$R/ab/libattn_qb2.so timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -6 > $O/attn_tests_qb2.txt
echo "[attention tests on the variant done $(( $(date +%s) - t0 )) s]" >> $O/attn_tests_qb2.txt
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['p50_latency_ms'])
except Exception as e: print('$1 FAILED', e)"; }
for r in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>$O/err.txt | line "product" >> $O/bench_ab.txt
  UNIDEPTH_HIP_LIB=$R/ab/libattn_qb2.so timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>$O/err.txt | line "qb2" >> $O/bench_ab.txt
done
echo "[bench ab done $(( $(date +%s) - t0 )) s]" >> $O/bench_ab.txt
cat $O/fixed_tests.txt $O/attn_ab.txt $O/attn_tests_qb2.txt $O/bench_ab.txt
}

# round 4, GPU call 10: the whole GPU suite on the current tree (timing per file), smoke()
call10() {
O=gpurun_out/r4c10 && mkdir -p $O
t0=$(date +%s)
timeout 1500 python -m pytest tests/ -q -m gpu --durations=25 2>&1 | grep -v "^$\|amdgpu.ids" | tail -60 > $O/suite.txt
echo "[suite done $(( $(date +%s) - t0 )) s]" >> $O/suite.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | grep -v amdgpu.ids | tail -4 > $O/smoke.txt
cat $O/suite.txt $O/smoke.txt
}

# round 4, GPU call 11: operand-power probe (attention + the encoder GEMMs on random / constant / zero operands), stream priority and
# three calls in flight on the bench line, interleaved
call11() {
O=gpurun_out/r4c11 && mkdir -p $O
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
}

# round 4, GPU call 12: the decoder's camera branch on the program's side stream -- tests (bit identity eager / graph / taps / pipeline),
# the tap-level parity tests, and the bench line with and without it, interleaved (the one-call p50 is the number it is for)
call12() {
O=gpurun_out/r4c12 && mkdir -p $O
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
}

# round 4, GPU call 13: head convolution with its weights in registers (conv_head_regw_kernel) against the LDS-streamed form
call13() {
O=gpurun_out/r4c13 && mkdir -p $O
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
}

# round 4, GPU call 14: the end-of-round bench line again with the kernel-class labels that match this build's rocprofv3 names (traffic
# from profiles/r04_hbm_traffic.json), per-launch table
call14() {
O=gpurun_out/r4c14 && mkdir -p $O
timeout 900 python bench.py --dump-ops $O/ops_per_launch.tsv > $O/bench.json 2> $O/bench.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/r4c14/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "p50_latency_ms", "value_one_call")})
print(json.dumps(d["roofline"])[:700])
print(json.dumps(d.get("roofline_enc_attention_mlp"))[:500])
for k, v in d.get("configs", {}).items():
    print(k, v.get("value"), v.get("ms_per_step"), v.get("error"), json.dumps(v)[:300] if k == "latency_bs1" else "")
print(json.dumps(d.get("kernel_breakdown", {}))[:1500])
P
}

# round 4, GPU call 15: x2 up-sampling fused into the ConvTranspose accumulate (UdGemm.up_src) against the materialised form
call15() {
O=gpurun_out/r4c15 && mkdir -p $O
t0=$(date +%s)
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "d2s" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -8 > $O/tests.txt
timeout 500 python -m pytest tests/test_parity_gpu.py tests/test_infer_gpu.py -q -m gpu -k "taps or golden or headline or more_shapes or seams" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -8 >> $O/tests.txt
echo "[tests done $(( $(date +%s) - t0 )) s]" >> $O/tests.txt
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['p50_latency_ms'])
except Exception as e: print('$1 FAILED', e)"; }
B="python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing"
for r in 1 2 3; do
  UNIDEPTH_UPFUSE=0 timeout 300 $B 2>$O/err.txt | line "materialised" >> $O/bench_ab.txt
  timeout 300 $B 2>$O/err.txt | line "fused" >> $O/bench_ab.txt
done
echo "[bench ab done $(( $(date +%s) - t0 )) s]" >> $O/bench_ab.txt
cat $O/tests.txt $O/bench_ab.txt
}

# round 4, GPU call 16: head conv kernel with the padded halo pitch + one-group software pipeline of the fragment reads, against its first form
call16() {
O=gpurun_out/r4c16 && mkdir -p $O
t0=$(date +%s)
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "head" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -6 > $O/tests.txt
timeout 500 python -m pytest tests/test_infer_gpu.py tests/test_parity_gpu.py -q -m gpu -k "golden or headline or taps" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -6 >> $O/tests.txt
echo "[tests done $(( $(date +%s) - t0 )) s]" >> $O/tests.txt
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['p50_latency_ms'])
except Exception as e: print('$1 FAILED', e)"; }
B="python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing"
for r in 1 2 3; do
  UNIDEPTH_HIP_LIB=$R/ab/libhead_prev.so timeout 300 $B 2>$O/err.txt | line "first_form" >> $O/bench_ab.txt
  timeout 300 $B 2>$O/err.txt | line "padded_pipelined" >> $O/bench_ab.txt
done
echo "[bench ab done $(( $(date +%s) - t0 )) s]" >> $O/bench_ab.txt
cat $O/tests.txt $O/bench_ab.txt
}

# round 4, end-of-round GPU call: rocprofv3 kernel stats + FETCH / WRITE passes (tools/profile_bench.sh), PMC passes over the encoder GEMMs and
# the attention kernel, the bench line with every sub-record and the per-launch table, the whole GPU suite, smoke()
final() {
O=gpurun_out/r4final && mkdir -p $O
t0=$(date +%s)
timeout 600 bash tools/profile_bench.sh r04 > $O/profile_bench.log 2>&1
python tools/update_profiles.py r04 r04_bench_bs8_vitl >> $O/profile_bench.log 2>&1
echo "[profiles done $(( $(date +%s) - t0 )) s]"
timeout 300 bash tools/pmc_gemm.sh 2>&1 | grep -v amdgpu.ids > $O/gemm_pmc.txt
timeout 200 bash tools/pmc_attn.sh 2>&1 | grep -v amdgpu.ids > $O/attn_pmc.txt
echo "[pmc done $(( $(date +%s) - t0 )) s]"
timeout 900 python bench.py --dump-ops $O/ops_per_launch.tsv > $O/bench.json 2> $O/bench.err
echo "[bench done $(( $(date +%s) - t0 )) s]"
mkdir -p $O/profiles && cp profiles/r04_bench_bs8_vitl_kernel_stats.csv profiles/r04_hbm_traffic.json profiles/r04_v1_* $O/profiles/ 2>/dev/null
rm -rf gpurun_out/prof_r04/*/ gpurun_out/pmcg_* gpurun_out/pmca_*          # raw traces stay on the box
timeout 1200 python -m pytest tests/ -q -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" | tail -15 > $O/suite.txt
echo "[suite done $(( $(date +%s) - t0 )) s]" >> $O/suite.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | grep -v amdgpu.ids | tail -4 > $O/smoke.txt
tail -4 $O/profile_bench.log; head -12 $O/gemm_pmc.txt; head -20 $O/attn_pmc.txt; cat $O/suite.txt $O/smoke.txt; tail -3 $O/bench.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/r4final/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "p50_latency_ms", "value_one_call")})
print(json.dumps(d["roofline"])[:900])
for k, v in d.get("configs", {}).items():
    print(k, v.get("value"), v.get("ms_per_step"), v.get("error"))
P
}

case "$1" in
  call1|call2|call3|call4|call5|call6|call7|call8|call9|call10|call11|call12|call13|call14|call15|call16|final) "$1" ;;
  *) echo "usage: $0 {call1|call2|call3|call4|call5|call6|call7|call8|call9|call10|call11|call12|call13|call14|call15|call16|final}"; exit 2 ;;
esac
