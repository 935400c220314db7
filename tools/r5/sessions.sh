#!/bin/bash
# The batched GPU-box sessions of round 5, one function per gpurun call (tests + interleaved A/B + profiles in one call each); every
# profiles/r05_* file names the session that produced it.  usage (on the GPU box, through gpurun):  bash tools/r5/sessions.sh <name>
# (function bodies are not indented: they contain here-documents)
cd "$(dirname "$0")/../.." && R=$PWD
export PYTHONWARNINGS=ignore
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('p50_latency_ms'), d.get('value_one_call'))
except Exception as e: print('$1 FAILED', e)"; }

# round 5, GPU call 1: the software-pipelined attention kernel (correctness incl. the spiked-key rescale path, interleaved A/B of its
# compile-time variants against the round-4 kernel, the attention kernel tests on the product library), the V2 parity sweep with the
# reference-as-shipped (fp16 autocast) comparator, the bench line with the new and the round-4 attention kernel interleaved
call1() {
O=gpurun_out/r5c1 && mkdir -p $O
t0=$(date +%s)
timeout 600 python tools/attn_ab.py --rounds 2 classic pipe o2fd4 o1 o0fd4 nw8 2>&1 | grep -v amdgpu.ids > $O/attn_ab.txt
echo "[ab done $(( $(date +%s) - t0 )) s]" >> $O/attn_ab.txt
timeout 400 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -8 > $O/attn_tests.txt
echo "[attention tests done $(( $(date +%s) - t0 )) s]" >> $O/attn_tests.txt
for r in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>$O/err.txt | line "pipe" >> $O/bench_ab.txt
  UNIDEPTH_HIP_LIB=$R/ab/libattn_classic.so timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>$O/err.txt | line "classic" >> $O/bench_ab.txt
done
echo "[bench ab done $(( $(date +%s) - t0 )) s]" >> $O/bench_ab.txt
timeout 900 python -m pytest tests/test_parity_sweep_gpu.py -q -s -m gpu -k "v2" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -110 > $O/sweep_v2.txt
echo "[v2 sweep done $(( $(date +%s) - t0 )) s]" >> $O/sweep_v2.txt
cat $O/attn_ab.txt $O/attn_tests.txt $O/bench_ab.txt; tail -70 $O/sweep_v2.txt
}

# round 5, GPU call 2: where does the pipelined attention kernel's tile time go?  Parts compiled out (results wrong by construction), plus two
# structural variants (fragment ring 4 deep; 2-wave workgroups = half the barrier group)
call2() {
O=gpurun_out/r5c2 && mkdir -p $O
timeout 900 python tools/attn_ab.py --rounds 2 classic pipe fd4 nw2 abl1 abl3 abl4 abl8 abl16 abl24 abl32 abl35 abl64 abl127 2>&1 | grep -v amdgpu.ids > $O/attn_abl.txt
cat $O/attn_abl.txt
}

# round 5, GPU call 3: the one-tile-at-a-time kernel with the cheap VALU savings of the pipelined one (c1 permlane exchange, c2 matrix-pipe row
# sums, c3 both), and the pipelined kernel ROLLED (one tile per trip: 149-173 registers instead of 252 -> three waves per SIMD)
call3() {
O=gpurun_out/r5c3 && mkdir -p $O
timeout 900 python tools/attn_ab.py --rounds 2 classic pipe c1 c2 c3 roll r3o2 r3o3 r3o0 2>&1 | grep -v amdgpu.ids > $O/attn_ab.txt
cat $O/attn_ab.txt
}

# round 5, GPU call 4: UniDepthV1 with the NystromBlocks as the reference executes them (UD_V1_HEAD_MIX): kernel + model tests, the V1 parity sweeps,
# the V1 bench at BASELINE configs[3]; the attention kernel tests again (split-key mode removed)
call4() {
O=gpurun_out/r5c4 && mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest tests/test_v1_gpu.py -q -m gpu -x 2>&1 | grep -v "^$\|amdgpu.ids" | tail -15 > $O/v1_tests.txt
echo "[v1 tests done $(( $(date +%s) - t0 )) s]" >> $O/v1_tests.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -5 > $O/attn_tests.txt
timeout 900 python -m pytest tests/test_parity_sweep_gpu.py -q -s -m gpu -k "v1" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -80 > $O/sweep_v1.txt
echo "[v1 sweeps done $(( $(date +%s) - t0 )) s]" >> $O/sweep_v1.txt
timeout 400 python tools/bench_v1.py > $O/bench_v1.txt 2>&1
timeout 400 python tools/bench_v1.py --by-tag > $O/bench_v1_tags.txt 2>&1
cat $O/v1_tests.txt $O/attn_tests.txt; tail -60 $O/sweep_v1.txt; tail -12 $O/bench_v1.txt; tail -45 $O/bench_v1_tags.txt
}

# round 5, GPU call 5: one infer() as two (four) sub-batches in flight (model.latency_split): bit-identity tests, the one-call latency with and
# without it on the headline workload
call5() {
O=gpurun_out/r5c5 && mkdir -p $O
timeout 600 python -m pytest tests/test_infer_gpu.py -q -m gpu -k "latency_split or pipeline or headline" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -8 > $O/tests.txt
lat() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value', d['value'], 'p50', d['p50_latency_ms'], 'p90', d['p90_latency_ms'], 'unsplit p50', d.get('p50_latency_ms_unsplit'), 'split', d.get('latency_split'))
except Exception as e: print('$1 FAILED', e)"; }
for r in 1 2; do
  for sp in 1 2 4; do
    timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing --latency-split $sp 2>$O/err.txt | lat "split=$sp" >> $O/latency_ab.txt
  done
done
cat $O/tests.txt $O/latency_ab.txt; tail -5 $O/err.txt
}

# round 5, GPU call 6: the whole GPU suite after the clean-up (side branch / hipGraph replay / env switches / split-key attention removed,
# ud_rccl_* added), a bench line, two bs = 4 requests in flight against one (how much do half-size programs overlap?)
call6() {
O=gpurun_out/r5c6 && mkdir -p $O
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_parity_sweep_gpu.py 2>&1 | grep -v "^$\|amdgpu.ids" | tail -15 > $O/suite.txt
echo "[suite done $(( $(date +%s) - t0 )) s]" >> $O/suite.txt
timeout 400 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs 2>$O/err.txt | line "bs8 inflight2" > $O/bench.txt
for inf in 1 2; do
  timeout 300 python bench.py --batch 4 --inflight $inf --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>>$O/err.txt | line "bs4 inflight$inf" >> $O/bench.txt
done
cat $O/suite.txt $O/bench.txt; tail -3 $O/err.txt
}

# round 5, GPU call 7: the GPU suite again (INTEGRATION.md snippet fixed; BatchCamera with one model per image), without -x
call7() {
O=gpurun_out/r5c7 && mkdir -p $O
t0=$(date +%s)
timeout 1800 python -m pytest tests -q -m gpu --deselect tests/test_parity_sweep_gpu.py 2>&1 | grep -v "^$\|amdgpu.ids" | tail -25 > $O/suite.txt
echo "[suite done $(( $(date +%s) - t0 )) s]" >> $O/suite.txt
cat $O/suite.txt
}

# round 5, GPU call 8: proj / fc2 residual preload behind a COUNTED operand wait (product) against the round-4 vmcnt(0) form (ab/libresid0.so):
# the GEMM kernel tests on the product, the four encoder GEMM shapes and the bench line interleaved
call8() {
O=gpurun_out/r5c8 && mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -5 > $O/gemm_tests.txt
for r in 1 2; do
  for lib in ab/libresid0.so unidepth_amd/libunidepth_hip.so; do
    echo "== $lib" >> $O/enc_gemms.txt
    UNIDEPTH_HIP_LIB=$R/$lib timeout 200 python tools/bench_enc_gemms.py 2>&1 | grep -v amdgpu.ids | tail -6 >> $O/enc_gemms.txt
    UNIDEPTH_HIP_LIB=$R/$lib timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>$O/err.txt | line "$lib" >> $O/bench_ab.txt
  done
done
cat $O/gemm_tests.txt $O/enc_gemms.txt $O/bench_ab.txt
}

# round 5, GPU call 9: PMC passes over the attention micro-benchmark for the pipelined kernel (product) and the one-tile-at-a-time kernel
# (ab/libattn_classic.so): cycles, instruction counts, MFMA / VALU busy, and the clock (GRBM_GUI_ACTIVE / kernel duration).  Counters only.
call9() {
O=$R/gpurun_out/r5c9 && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for arm in pipe classic; do
  lib=$R/unidepth_amd/libunidepth_hip.so; [ $arm = classic ] && lib=$R/ab/libattn_classic.so
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES" \
             "GRBM_GUI_ACTIVE" ; do
    i=$((i+1))
    UNIDEPTH_HIP_LIB=$lib timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/${arm}_$i -o p -- python $R/tools/bench_attn.py > $O/${arm}_$i.log 2>&1
  done
done
python - <<'PY' > $O/attn_pmc.txt
import csv, glob, collections, os
O = os.environ.get("O", "/root/repo/gpurun_out/r5c9")
for arm in ("pipe", "classic"):
    acc = collections.defaultdict(float); n = collections.defaultdict(int); dur = []
    for f in sorted(glob.glob(f"{O}/{arm}_[0-9]*/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if "attention" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for f in sorted(glob.glob(f"{O}/{arm}_3/**/*kernel_trace.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if "attention" in r["Kernel_Name"]:
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f"== {arm}: per-launch averages over the profiled launches of tools/bench_attn.py (encoder attention, B=8 H=16 N=1370, q pre-scaled)")
    for k in sorted(acc):
        print(f"{k:32s} {acc[k] / max(n[k], 1):16.0f}   (n={n[k]})")
    if dur and "GRBM_GUI_ACTIVE" in acc:
        d = sum(dur) / len(dur)
        print(f"kernel duration (GRBM pass)      {d / 1e3:16.1f} us   clock = GRBM_GUI_ACTIVE / duration = {acc['GRBM_GUI_ACTIVE'] / n['GRBM_GUI_ACTIVE'] / d:.3f} GHz")
PY
cd $R
rm -rf $O/pipe_[0-9] $O/classic_[0-9]
cat $O/attn_pmc.txt; tail -2 $O/pipe_1.log
}

# round 5, GPU call 10: the pipelined attention kernel with PERSISTENT workgroups (512 walk the 1408 items, the next item's Q / K(0) / V(0) / K(1)
# fetched under the last tile) against one workgroup per item and against the one-tile-at-a-time kernel; attention kernel tests; bench A/B
call10() {
O=gpurun_out/r5c10 && mkdir -p $O
timeout 600 python tools/attn_ab.py --rounds 3 classic nopersist persist 2>&1 | grep -v amdgpu.ids > $O/attn_ab.txt
timeout 400 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -6 > $O/attn_tests.txt
for r in 1 2; do
  for lib in ab/libattn_nopersist.so unidepth_amd/libunidepth_hip.so; do
    UNIDEPTH_HIP_LIB=$R/$lib timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>$O/err.txt | line "$lib" >> $O/bench_ab.txt
  done
done
cat $O/attn_ab.txt $O/attn_tests.txt $O/bench_ab.txt
}

# round 5, the end-of-round run: rocprofv3 kernel stats + FETCH / WRITE passes over bench.py (and the V1 trace), GEMM PMC passes, the full bench line with
# per-launch timings, the WHOLE GPU suite (incl. the parity sweeps), smoke()
final() {
O=gpurun_out/r5final && mkdir -p $O
t0=$(date +%s)
timeout 700 bash tools/profile_bench.sh r05 > $O/profile_bench.log 2>&1
python tools/update_profiles.py r05 r05_bench_bs8_vitl >> $O/profile_bench.log 2>&1
echo "[profiles done $(( $(date +%s) - t0 )) s]"
timeout 300 bash tools/pmc_gemm.sh 2>&1 | grep -v amdgpu.ids > $O/gemm_pmc.txt
echo "[pmc done $(( $(date +%s) - t0 )) s]"
timeout 900 python bench.py --dump-ops $O/ops_per_launch.tsv > $O/bench.json 2> $O/bench.err
echo "[bench done $(( $(date +%s) - t0 )) s]"
mkdir -p $O/profiles && cp profiles/r05_bench_bs8_vitl_kernel_stats.csv profiles/r05_hbm_traffic.json profiles/r05_v1_cnvnxtl_640x480_bs16_kernel_stats.csv $O/profiles/ 2>/dev/null
cp gpurun_out/prof_r05.v1.log $O/profiles/r05_v1_trace_breakdown.txt 2>/dev/null
rm -rf gpurun_out/prof_r05/*/ gpurun_out/pmcg_* gpurun_out/pmca_*          # raw traces stay on the box
timeout 1800 python -m pytest tests/ -q -s -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" > $O/suite_full.txt
tail -15 $O/suite_full.txt > $O/suite.txt
echo "[suite done $(( $(date +%s) - t0 )) s]" >> $O/suite.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | grep -v amdgpu.ids | tail -4 > $O/smoke.txt
tail -4 $O/profile_bench.log; head -12 $O/gemm_pmc.txt; cat $O/suite.txt $O/smoke.txt; tail -3 $O/bench.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/r5final/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "p50_latency_ms", "value_one_call")})
print(json.dumps(d["roofline"])[:900])
print(json.dumps(d.get("cpu_baseline"))[:400])
for k, v in d.get("configs", {}).items():
    print(k, v.get("value"), v.get("ms_per_step"), v.get("error"))
P
}

# what the driver runs at the end of the round, in its order and with its flags, on a fresh box
rehearsal() {
O=gpurun_out/r5rehearsal && mkdir -p $O
t0=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" | tail -8 > $O/suite.txt
echo "[suite $(( $(date +%s) - t0 )) s]" >> $O/suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | grep -v amdgpu.ids | tail -3 > $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "[all $(( $(date +%s) - t0 )) s]" >> $O/suite.txt
cat $O/suite.txt $O/smoke.txt; tail -2 $O/bench.err; cut -c1-400 $O/bench.json
}

# XCD row-owner tile map of the single-round large-tile launches (ab/libown.so) against the product library: tests, bench A/B (interleaved),
# FETCH_SIZE pass of both
call11() {
O=$R/gpurun_out/r5c11 && mkdir -p $O
ARMS="${ARMS:-base own rm}"
libof() { if [ $1 = base ]; then echo $R/unidepth_amd/libunidepth_hip.so; else echo $R/ab/lib$1.so; fi; }
for arm in $ARMS; do
  [ $arm = base ] && continue
  UNIDEPTH_HIP_LIB=$(libof $arm) timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_infer_gpu.py tests/test_parity_gpu.py -q -x -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" | tail -3 > $O/tests_$arm.txt
done
for rep in 1 2 3; do
  for arm in $ARMS; do
    UNIDEPTH_HIP_LIB=$(libof $arm) timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --dump-ops $O/ops_${arm}_$rep.tsv > $O/bench_${arm}_$rep.json 2> $O/bench_${arm}_$rep.err
  done
done
( cd /tmp && export TMPDIR=/tmp
  ARGS="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extra-configs --inflight 1"
  for arm in $ARMS; do
    UNIDEPTH_HIP_LIB=$(libof $arm) timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f_$arm -o f -- python $ARGS > $O/f_$arm.log 2>&1
  done )
ARMS="$ARMS" python - <<'P' | tee $O/ab.txt
import collections, csv, glob, json, os
O = "gpurun_out/r5c11"
for arm in os.environ["ARMS"].split():
    vals, p50s = [], []
    t = collections.defaultdict(list)
    for rep in (1, 2, 3):
        try:
            d = json.loads(open(f"{O}/bench_{arm}_{rep}.json").read().strip().splitlines()[-1])
            vals.append(d["value"]); p50s.append(d["p50_latency_ms"])
            for line in open(f"{O}/ops_{arm}_{rep}.tsv"):
                c = line.rstrip("\n").split("\t")
                if len(c) >= 4 and c[2] in ("enc.qkv", "enc.attn", "enc.proj", "enc.fc1", "enc.fc2", "patch.w"):
                    t[c[2]].append(float(c[3]))
        except Exception as e:
            print(arm, rep, "bench failed", e)
    print(arm, "value", vals, "p50", p50s)
    print(arm, {k: round(sum(v) / len(v), 2) for k, v in t.items()})
    a = collections.defaultdict(list)
    for path in glob.glob(f"{O}/f_{arm}/**/f_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            a[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(a.items()):
        if "gemm256" in k:
            print(f"  {arm} read MB/launch {sum(v) / len(v) * 2048 / 1e6:8.1f}  x{len(v):4d}  {k[28:90]}")
P
for arm in $ARMS; do rm -rf $O/f_$arm; done
cat $O/tests_*.txt
}

# the camera head as one launch (UdCameraHead): kernel tests, model tests, A/B against the per-layer launches (UNIDEPTH_CAMHEAD=0, a switch
# that exists for this session only)
call12() {
O=$R/gpurun_out/r5c12 && mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -m gpu -k "camera_head" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -15 > $O/tests_kernel.txt
if ! grep -q "passed" $O/tests_kernel.txt || grep -q "failed" $O/tests_kernel.txt; then cat $O/tests_kernel.txt; echo "kernel tests failed: stopping"; return; fi
timeout 900 python -m pytest tests/test_infer_gpu.py tests/test_parity_gpu.py -q -x -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" | tail -6 > $O/tests_model.txt
for rep in 1 2; do
  for arm in 1 0; do
    UNIDEPTH_CAMHEAD=$arm timeout 600 python bench.py --no-cpu-baseline --no-extra-configs --no-kernel-timing > $O/bench_${arm}_$rep.json 2> $O/bench_${arm}_$rep.err
    UNIDEPTH_CAMHEAD=$arm timeout 600 python bench.py --batch 1 --no-cpu-baseline --no-extra-configs --no-kernel-timing > $O/bench1_${arm}_$rep.json 2>> $O/bench_${arm}_$rep.err
  done
done
( cd /tmp && export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extra-configs --inflight 1 > $O/trace.log 2>&1 )
grep -h "camera_head\|linear_f32\|attention_small" $O/trace/*/t_kernel_stats.csv $O/trace/t_kernel_stats.csv 2>/dev/null | cut -c1-200 > $O/camera_kernels.txt
rm -rf $O/trace
python - <<'P' | tee $O/ab.txt
import json
O = "gpurun_out/r5c12"
for arm in ("1", "0"):
    for name in ("bench", "bench1"):
        v = []
        for rep in (1, 2):
            try:
                d = json.loads(open(f"{O}/{name}_{arm}_{rep}.json").read().strip().splitlines()[-1])
                v.append((d["value"], d["p50_latency_ms"]))
            except Exception as e:
                v.append(("failed", str(e)[:80]))
        print("one launch" if arm == "1" else "per layer ", "bs=8" if name == "bench" else "bs=1", "(images/s two in flight, p50 ms one call):", v)
P
cat $O/tests_kernel.txt $O/tests_model.txt $O/camera_kernels.txt
}

# UniDepthV1: the one-channel output convs as an fp32 stencil (UD_V1_OUT_CONV3): kernel test, V1 tests, parity sweep (V1 cases), bench + trace
call13() {
O=$R/gpurun_out/r5c13 && mkdir -p $O
timeout 300 python -m pytest tests/test_v1_gpu.py -q -x -m gpu -k "out_conv3" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -8 > $O/tests_kernel.txt
if ! grep -q "passed" $O/tests_kernel.txt || grep -q "failed" $O/tests_kernel.txt; then cat $O/tests_kernel.txt; echo "kernel tests failed: stopping"; return; fi
timeout 900 python -m pytest tests/test_v1_gpu.py -q -x -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" | tail -6 > $O/tests_v1.txt
timeout 900 python -m pytest tests/test_parity_sweep_gpu.py -q -s -m gpu -k "v1" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -60 > $O/sweep_v1.txt
( cd /tmp && export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v1trace -o v -- python $R/tools/bench_v1.py 16 --no-cpu > $O/v1_breakdown.txt 2>&1 )
cp $O/v1trace/*/v_kernel_stats.csv $O/v1_kernel_stats.csv 2>/dev/null || cp $O/v1trace/v_kernel_stats.csv $O/v1_kernel_stats.csv 2>/dev/null
rm -rf $O/v1trace
cat $O/tests_kernel.txt $O/tests_v1.txt; tail -25 $O/sweep_v1.txt; grep -v "^W2\|^E2\|^I2" $O/v1_breakdown.txt | cut -c1-200 | head -40
}

# the default bench line alone (another box of the pool)
benchonly() {
O=gpurun_out/r5bench && mkdir -p $O
timeout 900 python bench.py --dump-ops $O/ops_per_launch.tsv > $O/bench.json 2> $O/bench.err
tail -2 $O/bench.err; cut -c1-330 $O/bench.json
}

# FETCH_SIZE / WRITE_SIZE against known byte counts in the access patterns of the large-tile GEMM (tools/ubench/fetch_calib.hip)
calib() {
O=$PWD/gpurun_out/r5calib && mkdir -p $O
B=$PWD/tools/ubench/fetch_calib
( cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o f -- $B > $O/f.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o w -- $B > $O/w.log 2>&1 )
python - <<'P' | tee $O/calib.txt
import collections, csv, glob
def agg(pat):
    a = collections.defaultdict(list)
    for path in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(path)):
            a[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return a
f, w = agg("gpurun_out/r5calib/f/**/f_counter_collection.csv"), agg("gpurun_out/r5calib/w/**/w_counter_collection.csv")
B = 512 * 2 ** 20
print("kernel            FETCH_SIZE*1024 / bytes   WRITE_SIZE*1024 / bytes   (512 MiB touched once per launch)")
for k in sorted(set(f) | set(w)):
    fr = sum(f.get(k, [0])) / max(1, len(f.get(k, [0]))) * 1024 / B
    wr = sum(w.get(k, [0])) / max(1, len(w.get(k, [0]))) * 1024 / B
    print(f"{k:18s} {fr:10.3f} {wr:24.3f}")
P
rm -rf $O/f $O/w
}

"$@"
