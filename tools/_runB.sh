export MICO_GEMM_VARIANT=2
for v in w4n16 w4s16 w4s8 w4s12; do
  export MICO_HIP_LIB=$PWD/tools/probes/bin/libmico_$v.so
  echo "=== $v"
  timeout 300 python tools/gemm_bench.py --iters 5 2>&1 | grep -v amdgpu.ids
done
export MICO_HIP_LIB=$PWD/tools/probes/bin/libmico_w4s16.so
timeout 600 python -m pytest tests/test_gemm_large_gpu.py tests/test_gemm_split_gpu.py -q 2>&1 | tail -3
