#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box: kernel trace + stats, then HBM counters (separate passes, no sys-trace).
# usage: tools/profile_bench.sh <tag>     -> gpurun_out/prof_<tag>/{trace,fetch,write}
TAG=${1:-r01}
R=/root/repo
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extra-configs --inflight 1"      # one call at a time: clean per-kernel durations
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/trace -o t -- python $ARGS > $R/gpurun_out/prof_$TAG.trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG/fetch -o f -- python $ARGS > $R/gpurun_out/prof_$TAG.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG/write -o w -- python $ARGS > $R/gpurun_out/prof_$TAG.write.log 2>&1
# UniDepthV1 (ConvNeXt-L, 640x480, bs 16): kernel trace + stats only
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG/v1trace -o v -- python $R/tools/bench_v1.py 16 --no-cpu > $R/gpurun_out/prof_$TAG.v1.log 2>&1
ls -R $R/gpurun_out/prof_$TAG | head -30
