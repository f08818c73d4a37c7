#!/bin/bash
# tools/build_variant.sh <tag> <extra hipcc flags...>: an A/B build of libmico_hip.so with other flags for gemm.hip only ->
# tools/probes/bin/libmico_<tag>.so (select with MICO_HIP_LIB=...).  The other translation units come from mico_amd/csrc/build.
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p tools/probes/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result "$@" -c mico_amd/csrc/gemm.hip -o tools/probes/bin/gemm_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/probes/bin/gemm_$tag.o mico_amd/csrc/build/layernorm.o mico_amd/csrc/build/elementwise.o \
    mico_amd/csrc/build/attention.o mico_amd/csrc/build/loss.o mico_amd/csrc/build/swin.o -o tools/probes/bin/libmico_$tag.so
rm -f tools/probes/bin/gemm_$tag.o
