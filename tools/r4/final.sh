#!/bin/bash
# round 4, end-of-round GPU call: rocprofv3 kernel stats + FETCH / WRITE passes (tools/profile_bench.sh), PMC passes over the encoder GEMMs and
# the attention kernel, the bench line with every sub-record and the per-launch table, the whole GPU suite, smoke()
cd "$(dirname "$0")/../.." && R=$PWD && O=gpurun_out/r4final && mkdir -p $O
export PYTHONWARNINGS=ignore
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
