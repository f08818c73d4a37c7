#!/bin/bash
# On the GPU box: the microbench (tools/gemm_bench.py) for every variant library in tools/probes/bin, same shapes, one process each.
#   tools/probes/run_gemm_variants.sh "<gemm_bench args>" tag1 tag2 ...
args="$1"; shift
for tag in "$@"; do
  echo "=== $tag"
  MICO_HIP_LIB=tools/probes/bin/libmico_$tag.so python tools/gemm_bench.py $args 2>&1 | grep -E "^layer|TFLOP" | grep -v "^square" | awk '{ if ($1=="layer") print; else printf "%s %s %s  ", $1, $2, $(NF-1)} END {print ""}'
done
