#!/bin/bash
# round 4, GPU call 10: the whole GPU suite on the current tree (timing per file), smoke()
cd "$(dirname "$0")/../.." && R=$PWD && O=gpurun_out/r4c10 && mkdir -p $O
export PYTHONWARNINGS=ignore
t0=$(date +%s)
timeout 1500 python -m pytest tests/ -q -m gpu --durations=25 2>&1 | grep -v "^$\|amdgpu.ids" | tail -60 > $O/suite.txt
echo "[suite done $(( $(date +%s) - t0 )) s]" >> $O/suite.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | grep -v amdgpu.ids | tail -4 > $O/smoke.txt
cat $O/suite.txt $O/smoke.txt
