#!/bin/bash
# Builds libmico_hip variants that differ in -D flags of gemm.hip only (tools/probes/bin/libmico_<tag>.so), in parallel.
#   tools/probes/build_gemm_variants.sh tag1:"-DA=1 -DB=2" tag2:"-DC=3" ...
# Run a variant with MICO_HIP_LIB=tools/probes/bin/libmico_<tag>.so (mico_amd/_lib.py).
set -e
cd "$(dirname "$0")/../../mico_amd/csrc"
OUT=../../tools/probes/bin
mkdir -p $OUT
make -s -j8 >/dev/null
pids=()
for spec in "$@"; do
  tag="${spec%%:*}"; flags="${spec#*:}"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -c gemm.hip -o $OUT/gemm_$tag.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/gemm_$tag.o build/layernorm.o build/elementwise.o build/attention.o build/loss.o build/swin.o build/comm.o -ldl -o $OUT/libmico_$tag.so &&
    rm -f $OUT/gemm_$tag.o && echo "built $tag ($flags)" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
