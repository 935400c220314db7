#!/bin/bash
# round 3, GPU call 1: split-weight GEMM kernels + UniDepthV1 parity at the 1e-3 bar; cost of the split; SGPR delta of the K-wrap on the encoder GEMMs
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/r3c1 && O=gpurun_out/r3c1
export PYTHONWARNINGS=ignore
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "split or wrap or gemm" 2>&1 | tail -5 > $O/kernels.txt
timeout 900 python -m pytest tests/test_v1_gpu.py -x -q -m gpu -s 2>&1 | grep -v Warning | tail -40 > $O/v1.txt
for i in 1 2; do
  UNIDEPTH_HIP_LIB=$PWD/ab/libold.so UNIDEPTH_HIP_LIB_ALLOW_OLDER=1 timeout 300 python tools/bench_enc_gemms.py 2>&1 | tail -1 | sed 's/^/old /' >> $O/enc_gemms.txt
  timeout 300 python tools/bench_enc_gemms.py 2>&1 | tail -1 | sed 's/^/new /' >> $O/enc_gemms.txt
done
UNIDEPTH_V1_WSPLIT=1 timeout 600 python tools/bench_v1.py 16 --no-cpu > $O/v1_bench_split.txt 2>&1
UNIDEPTH_V1_WSPLIT=0 timeout 600 python tools/bench_v1.py 16 --no-cpu > $O/v1_bench_single.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.txt 2>&1
tail -3 $O/kernels.txt; tail -12 $O/v1.txt; cat $O/enc_gemms.txt; head -c 600 $O/v1_bench_split.txt; echo; head -c 600 $O/v1_bench_single.txt; echo; head -c 400 $O/bench.txt
