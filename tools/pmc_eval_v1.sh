#!/bin/bash
# rocprofv3 PMC passes (counters only) over the K-NN benchmark and the UniDepthV1 step: VALU / LDS / wait shares of knn1_d3_kernel and
# dwconv7_lds_kernel.  GPU box; output: gpurun_out/pmce_<n>/, summary on stdout.
cd /tmp && export TMPDIR=/tmp
R=/root/repo
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmce_k$i -o p -- python $R/tools/bench_eval_ops.py > $R/gpurun_out/pmce_k$i.log 2>&1
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmce_v$i -o p -- python $R/tools/bench_v1.py 16 > $R/gpurun_out/pmce_v$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections
for tag, pat in (("knn1_d3_kernel", "pmce_k"), ("knn_kernel<3, 4>", "pmce_k"), ("dwconv7_lds_kernel", "pmce_v"), ("bmm_small_kernel", "pmce_v"), ("softmax_rows_kernel", "pmce_v")):
    print("==", tag)
    for f in sorted(glob.glob(f"/root/repo/gpurun_out/{pat}[0-9]*/**/*counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if tag in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in sorted(acc):
            print(f"  {k:28s} per launch {acc[k] / max(n[k], 1):16.0f}   (n={n[k]})")
PY
