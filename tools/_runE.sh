mkdir -p gpurun_out/r2h
timeout 900 python -m pytest tests/test_gemm_large_gpu.py tests/test_host_cpu.py -q -x -k "colsum or split_k or ctypes" 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_droppath_gpu.py -q -x 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r2h/bench_colsum.json 2> gpurun_out/r2h/bench_colsum.err
python - <<PY
import json
d = json.load(open("gpurun_out/r2h/bench_colsum.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["all_gemm"], d["roofline"]["achieved"])
PY
