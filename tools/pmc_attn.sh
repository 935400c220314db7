#!/bin/bash
# rocprofv3 PMC passes over the attention micro-benchmark (GPU box; counters only: no sys/runtime trace).  Output: gpurun_out/pmca_<n>/
cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmca_$i -o p -- python $R/tools/bench_attn.py > $R/gpurun_out/pmca_$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("/root/repo/gpurun_out/pmca_[0-9]*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "attention_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in sorted(acc):
        print(f"{k:32s} per launch {acc[k] / max(n[k], 1):16.0f}   (n={n[k]})")
PY
