mkdir -p gpurun_out/r2g
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > gpurun_out/r2g/pytest.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r2g/bench_default.json 2> gpurun_out/r2g/bench_default.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-gemm-timer --no-extras --no-cpu-baseline > gpurun_out/r2g/bench_notimer.json 2> gpurun_out/r2g/bench_notimer.err
timeout 600 python bench.py --steps 10 --warmup 3 --dtype bf16 --no-extras --no-cpu-baseline > gpurun_out/r2g/bench_bf16.json 2> gpurun_out/r2g/bench_bf16.err
timeout 600 python bench.py --workload vid_cap_fp8 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2g/bench_fp8.json 2> gpurun_out/r2g/bench_fp8.err
timeout 600 python bench.py --workload vid_cap_fp8 --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2g/bench_vidcap_bf16.json 2> gpurun_out/r2g/bench_vidcap_bf16.err
cat gpurun_out/r2g/pytest.log; for f in default notimer bf16 fp8 vidcap_bf16; do echo "== $f"; tail -3 gpurun_out/r2g/bench_$f.err; python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r2g/bench_$f.json"))
    print({k: d[k] for k in ("value", "ms_per_step", "dtype", "peak_mem_gb", "step_mfma_frac") if k in d})
    for k in ("parity", "parity_config", "secondary", "cpu_baseline"):
        if k in d: print(k, json.dumps(d[k])[:600])
    if d.get("roofline"): print("roofline", d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["all_gemm"])
except Exception as e: print("ERR", e)
PY
done
