#!/bin/bash
# rocprofv3 PMC passes over the encoder-GEMM micro-benchmark (GPU box; counters only).  Prints per-kernel MFMA utilisation:
# SQ_VALU_MFMA_BUSY_CYCLES (summed over all SIMDs, 16 per v_mfma_f32_16x16x32_f16) / (GPU-active cycles x 1024 SIMDs), where
# GPU-active cycles = GRBM_GUI_ACTIVE / 8 (the counter is reported summed over the 8 XCDs), and the wave-cycle breakdown.
cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcg_$i -o p -- python $R/tools/bench_enc_gemms.py > $R/gpurun_out/pmcg_$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(__import__("os").environ.get("UD_PMC_DIR", "/root/repo/gpurun_out") + "/pmcg_[0-9]*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if "gemm256" not in k and "gemm_pp" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    a = {c: acc[k][c] / n[k][c] for c in acc[k]}
    cyc = a.get("GRBM_GUI_ACTIVE", 8) / 8.0
    util = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(cyc * 1024, 1)
    print(f"{k}: MFMA busy {100 * util:.1f} % of SIMD-cycles while the GPU is active ({cyc:.0f} GPU cycles per launch)")
    for c in sorted(a): print(f"    {c:30s} {a[c]:16.0f}")
PY
