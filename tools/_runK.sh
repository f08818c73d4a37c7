MICO_HIP_LIB=tools/probes/bin/libmico_pcbk64.so timeout 900 python -m pytest tests/test_gemm_large_gpu.py -q -x -k "weight" 2>&1 | tail -2
for m in 82240 65792; do
echo "== bk64 $m"; MICO_HIP_LIB=tools/probes/bin/libmico_pcbk64.so python tools/gemm_bench.py --dtype fp16 --only dw --m $m 2>&1 | grep -v Warn | tail -5
echo "== bk32 $m"; python tools/gemm_bench.py --dtype fp16 --only dw --m $m 2>&1 | grep -v Warn | tail -5
done
