#!/bin/bash
# round 3, GPU call 20 (end of round): the whole GPU suite, smoke(), the default bench line, rocprofv3 passes, PMC pass over the encoder GEMMs
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r3c20 && O=gpurun_out/r3c20
export PYTHONWARNINGS=ignore
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $O/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 1200 python bench.py --dump-ops $O/ops.tsv > $O/bench.txt 2> $O/bench.err
bash tools/profile_bench.sh r03 > $O/profile.log 2>&1
bash tools/pmc_gemm.sh > $O/pmc_gemm.txt 2>&1
cat $O/tests.txt; tail -2 $O/smoke.txt; head -c 600 $O/bench.txt; echo; tail -30 $O/pmc_gemm.txt | cut -c1-160
