echo "=== persistent (default)"; timeout 300 python tools/gemm_bench.py --iters 10 2>&1 | grep -v amdgpu.ids | grep -v " dw "
echo "=== one workgroup per tile (variant 4)"; MICO_GEMM_VARIANT=4 timeout 300 python tools/gemm_bench.py --iters 10 2>&1 | grep -v amdgpu.ids | grep -v " dw "
timeout 900 python -m pytest tests/test_gemm_large_gpu.py tests/test_kernels_gpu.py -q -k "gemm or large" 2>&1 | tail -3
