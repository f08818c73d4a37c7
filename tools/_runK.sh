timeout 900 python -m pytest tests/test_gemm_large_gpu.py tests/test_kernels_gpu.py -q -x -k "gemm or weight" 2>&1 | tail -2
python tools/gemm_bench.py --dtype fp16 --only dw 2>&1 | grep -v Warn | tail -5
python tools/gemm_bench.py --dtype fp16 --only dw --m 65792 2>&1 | grep -v Warn | tail -5
