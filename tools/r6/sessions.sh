#!/bin/bash
# The batched GPU-box sessions of round 6, one function per gpurun call; every profiles/r06_* file names the session that produced it.
# usage (on the GPU box, through gpurun):  bash tools/r6/sessions.sh <name>
cd "$(dirname "$0")/../.." && R=$PWD
export PYTHONWARNINGS=ignore
line() { python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('p50_latency_ms'), d.get('value_one_call'))
except Exception as e: print('$1 FAILED', e)"; }

# round 6, GPU call 1: the product's K loop on plain 4096^3 / 8192^3 / encoder-shape problems against the ping-pong
# (staggered 8-phase, s_setprio) form of the same tile (tools/ubench/gemm8p.hip) on the same operand fill and box; the pure-MFMA stream
# of tools/ubench/gemm4w for the box's attainable rate; a bench line for the box's level
call1() {
O=gpurun_out/r6c1 && mkdir -p $O
t0=$(date +%s)
for s in "4096 4096 4096" "8192 8192 8192" "11008 1024 1024" "11008 3072 1024" "11008 4096 1024" "11008 1024 4096"; do
  timeout 120 tools/ubench/gemm8p $s 2>&1 | grep -v amdgpu.ids >> $O/gemm8p.txt
done
echo "[gemm8p done $(( $(date +%s) - t0 )) s]" >> $O/gemm8p.txt
timeout 300 python tools/bench_gemm_plain.py 2>&1 | grep -v amdgpu.ids > $O/product_plain.txt
echo "[product done $(( $(date +%s) - t0 )) s]" >> $O/product_plain.txt
timeout 120 tools/ubench/gemm4w 16384 4096 4096 2>&1 | grep -v amdgpu.ids > $O/gemm4w.txt
timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs 2>$O/err.txt | tee $O/bench.json | line "bench" > $O/bench.txt
cat $O/gemm8p.txt $O/product_plain.txt $O/gemm4w.txt $O/bench.txt; tail -3 $O/err.txt
}

# round 6, GPU call 2: the same A/B in ONE process (the product's kernel through its C-ABI from the micro-benchmark, arms interleaved: call 1
# showed the same product kernel 9 % apart between two places of one python loop -- clock ramp); --inflight 1 / 2 / 3 / 4 on the bench line
call2() {
O=gpurun_out/r6c2 && mkdir -p $O
for s in "4096 4096 4096" "11008 1024 1024" "11008 1024 4096" "11008 3072 1024" "11008 4096 1024"; do
  UD_LIB=$R/unidepth_amd/libunidepth_hip.so timeout 120 tools/ubench/gemm8p $s 2>&1 | grep -v amdgpu.ids >> $O/gemm8p_vs_product.txt
done
for r in 1 2; do for inf in 2 3 4; do
  timeout 300 python bench.py --steps 24 --warmup 4 --inflight $inf --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>$O/err.txt | line "inflight=$inf" >> $O/inflight.txt
done; done
cat $O/gemm8p_vs_product.txt $O/inflight.txt; tail -3 $O/err.txt
}

# round 6, GPU call 3: the two-way K split of the large-tile list (gemm256_kernel SPK) on the stage-0 RCU convolutions against the 128 x 128
# kernel they ran on and the unsplit 192-row list; the x4 fp32-accumulate launches as one large-tile list (re-measure of round 3's decision);
# the GEMM kernel tests
call3() {
O=gpurun_out/r6c3 && mkdir -p $O
timeout 600 python tools/r6_dec_ab.py --match dh.ups.0 --hints 1,3,10,0 2>&1 | grep -v amdgpu.ids > $O/stage0_split.txt
timeout 600 python tools/r6_dec_ab.py --match "dec.adapters,dh.out,dh.fc2" --hints 0,2 2>&1 | grep -v amdgpu.ids > $O/x4_big.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -8 > $O/gemm_tests.txt
timeout 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>$O/err.txt | line "bench" > $O/bench.txt
cat $O/stage0_split.txt $O/x4_big.txt $O/gemm_tests.txt $O/bench.txt; tail -3 $O/err.txt
}

# round 6, GPU call 4: ConvT depth-to-space launches and the stage-1 / stage-2 RCU convolutions per tile schedule
call4() {
O=gpurun_out/r6c4 && mkdir -p $O
timeout 600 python tools/r6_dec_ab.py --match dh.convt --hints 0,1,2,3 2>&1 | grep -v amdgpu.ids > $O/d2s.txt
timeout 600 python tools/r6_dec_ab.py --match "dh.ups.1,dh.ups.2,dh.to_latents" --hints 0,2,3 2>&1 | grep -v amdgpu.ids > $O/rcu.txt
timeout 600 python tools/r6_dec_ab.py --match "dec.adapters" --hints 0,1 2>&1 | grep -v amdgpu.ids > $O/adapters.txt
cat $O/d2s.txt $O/rcu.txt $O/adapters.txt
}

# round 6, GPU call 5: the ping-pong kernel (csrc/gemm_pp.hip) on the fp32 residual-accumulate class: bit identity with the 192-row list, the
# K-split test, the four encoder GEMM launches with their real epilogues per schedule (UD_TILE_HINTS: 3 = 192-row list for proj / fc2, 11 =
# ping-pong; qkv / fc1 ignore 11), interleaved, and the bench line against the previous library (ab/libprev.so = HEAD~ build, if present)
call5() {
O=gpurun_out/r6c5 && mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "ping_pong or k_split or gemm" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -12 > $O/tests.txt
UD_TILE_HINTS=3,11,0 timeout 300 python tools/bench_enc_gemms.py 2>&1 | grep -v amdgpu.ids > $O/enc_gemms.txt
UD_TILE_HINTS=11,3,0 timeout 300 python tools/bench_enc_gemms.py 2>&1 | grep -v amdgpu.ids >> $O/enc_gemms.txt
for r in 1 2 3; do
  timeout 300 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>$O/err.txt | line "new" >> $O/bench_ab.txt
  [ -f ab/libprev.so ] && UNIDEPTH_HIP_LIB=$R/ab/libprev.so timeout 300 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extra-configs --no-kernel-timing 2>$O/err.txt | line "prev" >> $O/bench_ab.txt
done
cat $O/tests.txt $O/enc_gemms.txt $O/bench_ab.txt; tail -3 $O/err.txt
}

# round 6, GPU call 6: the whole GPU suite on the tree with the ping-pong proj / fc2 kernel, the K split of the stage-0 convolutions, the loud
# camera-head time-out + serialised camera-head launches; the full bench line (live MFMA calibration, reference-as-shipped leg, per-launch table)
call6() {
O=gpurun_out/r6c6 && mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^$\|amdgpu.ids" | tail -15 > $O/tests.txt
timeout 900 python bench.py --dump-ops $O/ops_per_launch.tsv 2>$O/err.txt > $O/bench.json
line "bench" < $O/bench.json > $O/bench.txt
cat $O/tests.txt $O/bench.txt; tail -3 $O/err.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6c6/bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print({k:r[k] for k in ('kernel','achieved','frac','attainable_this_box','frac_of_attainable','avg_launch_us','scope_frac','scope_ms_per_step')})
print(d['configs'].get('reference_as_shipped_rocm')); print(d['configs']['v1_cnvnxtl_640x480_bs16'].get('value'), d['configs']['latency_bs1'])
PY
}

# round 6, GPU call 7: the camera-head failure tests, the C-ABI exchange route, the software-pipelined depth-to-space epilogue (D2S kernel tests, the
# three ConvT launches new / previous library, same box), cost of the camera-head ordering events (cam.head row of the per-launch table, new / prev),
# bench A/B (ab/libprev.so = this tree built with -DUD_AB_PREV: round-6 schedules and the ordering events off, D2S epilogue as before is NOT part of it)
call7() {
O=gpurun_out/r6c7 && mkdir -p $O
timeout 900 python -m pytest tests/test_infer_gpu.py tests/test_rccl_gpu.py tests/test_kernels_gpu.py -q -m gpu -x -k "camera or cabi or dist_module or d2s or ping_pong or k_split" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -12 > $O/tests.txt
timeout 600 python tools/r6_dec_ab.py --match dh.convt --hints 0 --rounds 6 2>&1 | grep -v amdgpu.ids > $O/d2s_new.txt
[ -f ab/libd2sprev.so ] && UNIDEPTH_HIP_LIB=$R/ab/libd2sprev.so timeout 600 python tools/r6_dec_ab.py --match dh.convt --hints 0 --rounds 6 2>&1 | grep -v amdgpu.ids > $O/d2s_prev.txt
for r in 1 2; do
  timeout 300 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extra-configs --dump-ops $O/ops_new$r.tsv 2>$O/err.txt | line "new" >> $O/bench_ab.txt
  [ -f ab/libd2sprev.so ] && UNIDEPTH_HIP_LIB=$R/ab/libd2sprev.so timeout 300 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extra-configs --dump-ops $O/ops_d2sprev$r.tsv 2>$O/err.txt | line "d2sprev" >> $O/bench_ab.txt
  [ -f ab/libprev.so ] && UNIDEPTH_HIP_LIB=$R/ab/libprev.so timeout 300 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extra-configs --dump-ops $O/ops_prev$r.tsv 2>$O/err.txt | line "prev" >> $O/bench_ab.txt
done
grep -h "cam.head\|convt" $O/ops_*.tsv | cut -f3,4 > $O/rows.txt
cat $O/tests.txt $O/d2s_new.txt $O/d2s_prev.txt $O/bench_ab.txt; tail -3 $O/err.txt; paste - - - - < $O/rows.txt | head -8
}

# round 6, GPU call 8: the camera branch as a side section of the launch program: bit identity + cut replays (tests), A/B against the single-stream
# program in one process at bs 8 and bs 1 (one call p50, two in flight), the infer / parity tests on the forked plan, the bench line
call8() {
O=gpurun_out/r6c8 && mkdir -p $O
timeout 1200 python -m pytest tests/test_infer_gpu.py tests/test_parity_gpu.py -q -m gpu -x 2>&1 | grep -v "^$\|amdgpu.ids" | tail -8 > $O/tests.txt
# (tools/r6_fork_ab.py and the side-section ops exist in commits 5c0... "launch programs: side sections" .. the one after; removed when the A/B came back negative)
[ -f tools/r6_fork_ab.py ] && timeout 600 python tools/r6_fork_ab.py --batch 8 --rounds 5 2>&1 | grep -v amdgpu.ids > $O/fork_ab.txt
[ -f tools/r6_fork_ab.py ] && timeout 600 python tools/r6_fork_ab.py --batch 1 --rounds 5 --calls 60 2>&1 | grep -v amdgpu.ids >> $O/fork_ab.txt
timeout 300 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extra-configs 2>$O/err.txt | line "bench" > $O/bench.txt
cat $O/tests.txt $O/fork_ab.txt $O/bench.txt; tail -3 $O/err.txt
}

# round 6, final-tree evidence (one box): rocprofv3 kernel stats + FETCH / WRITE passes of the bench workload (profiles/r06_*), MFMA-busy counters of the
# encoder GEMM launches, the full bench line with the per-launch table, the whole GPU suite (with the parity sweep printed), smoke()
final() {
O=gpurun_out/r6final && mkdir -p $O
t0=$(date +%s)
timeout 700 bash tools/profile_bench.sh r06 > $O/profile_bench.log 2>&1
python tools/update_profiles.py r06 r06_bench_bs8_vitl >> $O/profile_bench.log 2>&1
echo "[profiles done $(( $(date +%s) - t0 )) s]"
timeout 300 bash tools/pmc_gemm.sh 2>&1 | grep -v amdgpu.ids > $O/gemm_pmc.txt
echo "[pmc done $(( $(date +%s) - t0 )) s]"
timeout 900 python bench.py --dump-ops $O/ops_per_launch.tsv > $O/bench.json 2> $O/bench.err
echo "[bench done $(( $(date +%s) - t0 )) s]"
mkdir -p $O/profiles && cp profiles/r06_bench_bs8_vitl_kernel_stats.csv profiles/r06_hbm_traffic.json profiles/r06_v1_cnvnxtl_640x480_bs16_kernel_stats.csv $O/profiles/ 2>/dev/null
cp gpurun_out/prof_r06.v1.log $O/profiles/r06_v1_trace_breakdown.txt 2>/dev/null
rm -rf gpurun_out/prof_r06/*/ gpurun_out/pmcg_* gpurun_out/pmca_*          # raw traces stay on the box
timeout 1800 python -m pytest tests/ -q -s -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" > $O/suite_full.txt
tail -15 $O/suite_full.txt > $O/suite.txt
echo "[suite done $(( $(date +%s) - t0 )) s]" >> $O/suite.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | grep -v amdgpu.ids | tail -4 > $O/smoke.txt
tail -4 $O/profile_bench.log; head -12 $O/gemm_pmc.txt; cat $O/suite.txt $O/smoke.txt; tail -3 $O/bench.err
python - <<'P'
import json
d = json.loads(open("gpurun_out/r6final/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "p50_latency_ms", "value_one_call")})
print(json.dumps(d["roofline"])[:1200])
print(json.dumps(d.get("cpu_baseline"))[:400])
for k, v in d.get("configs", {}).items():
    print(k, v.get("value"), v.get("ms_per_step"), v.get("p50_latency_ms"), v.get("error"))
P
}

# what the driver runs at the end of the round, in its order and with its flags, on a fresh box
rehearsal() {
O=gpurun_out/r6rehearsal && mkdir -p $O
t0=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" | tail -8 > $O/suite.txt
echo "[suite $(( $(date +%s) - t0 )) s]" >> $O/suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | grep -v amdgpu.ids | tail -3 > $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "[all $(( $(date +%s) - t0 )) s]" >> $O/suite.txt
cat $O/suite.txt $O/smoke.txt; tail -2 $O/bench.err; cut -c1-400 $O/bench.json
}

# round 6, GPU call 9: decoder token rows padded to a tile-aligned group height (1392 rows per image at bs 8: the x4 launches as 192-row tile lists,
# accumulating ones included) against the plain multiple of 8, one process; kernel tests of the grouped launches; infer / parity tests on the new plans
# (negative: profiles/r06_decoder_rows_ab.txt; the aligned rows, the 192-row grouped form and tools/r6_rows_ab.py were removed again)
call9() {
O=gpurun_out/r6c9 && mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "grouped" 2>&1 | grep -v "^$\|amdgpu.ids" | tail -6 > $O/tests.txt
[ -f tools/r6_rows_ab.py ] && timeout 900 python tools/r6_rows_ab.py --batch 8 --rounds 5 2>&1 | grep -v amdgpu.ids > $O/rows_ab.txt
timeout 1500 python -m pytest tests/test_infer_gpu.py tests/test_parity_gpu.py -q -m gpu -x 2>&1 | grep -v "^$\|amdgpu.ids" | tail -6 >> $O/tests.txt
timeout 300 python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extra-configs 2>$O/err.txt | line "bench" > $O/bench.txt
cat $O/tests.txt $O/rows_ab.txt $O/bench.txt; tail -3 $O/err.txt
}

# round 6, GPU calls 10 / 11: where the dispatcher puts 4-wave / 80 KB workgroups (tools/ubench/placement), then the
# two-workgroups-per-CU instantiation of the ping-pong kernel against the product's schedules, hot and with the caches flushed between launches
call10() {
O=gpurun_out/r6c10 && mkdir -p $O
for a in "464 81920" "512 81920" "464 65536" "464 81984"; do tools/ubench/placement $a 2>&1 | grep -v amdgpu.ids; done > $O/placement.txt
cat $O/placement.txt
}
call11() {
O=gpurun_out/r6c11 && mkdir -p $O
timeout 300 python tools/r6_duo_ab.py 2>&1 | grep -v amdgpu.ids > $O/duo_ab.txt
timeout 300 python tools/r6_duo_ab.py --cold 2>&1 | grep -v amdgpu.ids >> $O/duo_ab.txt
cat $O/duo_ab.txt
}

# round 6, GPU calls 12-28 (second half of the round), as run (one gpurun call each; outputs under gpurun_out/r6cNN, summaries under profiles/r06_*):
#  12  UD_TILE_HINTS=15,8,15,8,0 tools/bench_enc_gemms.py            rotated tall tiles of the balanced schedule on fc1 (no effect; removed)      -> r06_schedules_ab.txt
#  13  tools/r6_dec_ab.py --match dh.ups.{1,2}.*conv --hints 2,3,15,8,0   balanced schedule for zero-padded 3x3 operands                          -> r06_schedules_ab.txt
#  14  kernel + infer + parity tests, bench line with the balanced convs
#  15  bench_enc_gemms with / without epilogue stores (-DUD_EXP_NOSTORE build)                                                                      -> r06_schedules_ab.txt
#  16  UdLayerNorm.cls_y tests, infer / parity tests, bench line
#  17  tools/r6_conv2_epi.py                                           epilogue variants of the stage-2 conv2                                       -> r06_conv2_epilogue.txt
#  18  tools/r6_conv2_trace.py with ab/libtrace.so / libtraced.so (-DUD_TRACE [-DUD_TRACE_DRAIN])                                                   -> r06_conv2_epilogue.txt
#  19  the same with every other workgroup 36 us late (-DUD_EXP_DELAY=3500 experiment build)                                                        -> r06_conv2_epilogue.txt
#  20  tools/ubench/mfma_shape_power                                   pure MFMA streams per shape / operand fill                                   -> r06_kloop_ablation.txt
#  21, 22  UD_ABLATE=1 tools/ubench/gemm8p 4096^3, 16384 x 4096 x 4096   K-loop ablation ladder                                                     -> r06_kloop_ablation.txt
#  (final)  bash tools/r6/sessions.sh final                            kernel stats, FETCH / WRITE, PMC, bench + per-launch table, suite, smoke      -> r06_bench*.json, r06_*kernel_stats.csv, ...
#  23  UD_PH2=1 tools/ubench/gemm8p (three shapes)                     two phases per K-tile                                                         -> r06_kloop_ablation.txt
#  24  kernel tests, tools/r6_duo_ab.py --hints 3,11 and bench.py against ab/libpp4.so (four-phase build of HEAD~)                                  -> r06_schedules_ab.txt
#  25  UD_PH2=1 gemm8p with the one-phase variant (dropped)                                                                                         -> r06_kloop_ablation.txt
#  (rehearsal)  bash tools/r6/sessions.sh rehearsal                    driver order on a fresh box                                                  -> r06_bench_rehearsal.json, r06_rehearsal.txt
#  26-28  tests/test_v1_gpu.py, the V1 parity sweep, tools/bench_v1.py 16 --no-cpu [--by-tag] with UNIDEPTH_V1_DWLN=1 / 0 interleaved              -> r06_v1_dwconv_ln_ab.txt
#  (rehearsal, again on the last commit)                                                                                                            -> r06_bench_rehearsal2.json, r06_rehearsal2.txt

"$@"
