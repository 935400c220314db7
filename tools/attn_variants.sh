#!/bin/bash
# Build one library per attention-kernel variant (compile-time switches of csrc/attention.hip) into ab/libattn_<name>.so.
# Usage: tools/attn_variants.sh name1="flags" name2="flags" ...     (run in the authoring container; the .so files travel with gpurun)
set -e
cd "$(dirname "$0")/.."
mkdir -p ab
CS=unidepth_amd/csrc
[ -f $CS/build/gemm.o ] || bash $CS/build.sh
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  bd=/tmp/ud_attn_$name; mkdir -p $bd
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize $flags \
        -c $CS/attention.hip -o $bd/attention.o
  objs=$(ls $CS/build/*.o | grep -v attention.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs $bd/attention.o -o ab/libattn_$name.so
  echo "built ab/libattn_$name.so  ($flags)"
done
