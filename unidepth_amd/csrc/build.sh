#!/bin/bash
# Build libunidepth_hip.so for gfx950 in-tree (cross-compiles without a GPU).  Usage: csrc/build.sh [extra hipcc flags]
# UD_OUT / UD_BUILD_DIR override the output library / object directory (A/B and instrumented builds, tools/).
set -e
cd "$(dirname "$0")"
OUT=${UD_OUT:-../libunidepth_hip.so}
BD=${UD_BUILD_DIR:-build}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result"
mkdir -p $BD
pids=()
for f in gemm.hip gemm_pp.hip calib.hip layernorm.hip pointwise.hip camera_f32.hip convnext.hip v1dec.hip; do
  ( hipcc $FLAGS "$@" -c $f -o $BD/${f%.hip}.o ) &
  pids+=($!)
done
# attention: keep the MFMA accumulators in VGPRs (the softmax VALU works on them every tile; the default AGPR form costs
# ~160 v_accvgpr_read/write per 16 MFMAs); no SLP vectorisation: packed f32 adds (v_pk_add_f32) next to MFMAs are slower than
# the scalar adds they replace (half-rate issue, /opt/skills/guides MI355X_MICROARCH "price of one filler beside MFMAs")
( hipcc $FLAGS -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize "$@" -c attention.hip -o $BD/attention.o ) & pids+=($!)
# evaluation-side kernels: bit-exact fp32 distances (separately rounded multiply and add, like the reference's CPU build), so no FMA
# contraction anywhere in this file (the in-source pragma does not reach ext_vector_type arithmetic)
( hipcc $FLAGS -ffp-contract=off "$@" -c evalops.hip -o $BD/evalops.o ) & pids+=($!)
( hipcc $FLAGS -x hip -c api.cpp -o $BD/api.o ) & pids+=($!)
( hipcc $FLAGS -x hip -c program.cpp -o $BD/program.o ) & pids+=($!)
( hipcc $FLAGS -x hip -c rccl.cpp -o $BD/rccl.o ) & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC $BD/*.o -ldl -o $OUT
echo "built $(realpath $OUT)"
