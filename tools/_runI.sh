mkdir -p gpurun_out/i
MICO_PRECISION_STATS_N=6 MICO_PRECISION_STATS_CONFIGS=fp16,fp16-plain,fp16-split-w timeout 900 python -m pytest tests/test_precision_stats_gpu.py -q -x -s 2>&1 | grep -v Warning | tail -8
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_droppath_gpu.py tests/test_optim_gpu.py -q -x 2>&1 | tail -3
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/i/bench_head4.json 2> gpurun_out/i/bench_head4.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --dtype fp16-plain > gpurun_out/i/bench_plain.json 2> gpurun_out/i/bench_plain.err
python - <<PY
import json
for t in ("head4","plain"):
    d = json.load(open(f"gpurun_out/i/bench_{t}.json"))
    print(t, d["value"], d["ms_per_step"], d.get("parity"), d["roofline"]["all_gemm"])
PY
