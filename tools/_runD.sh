mkdir -p gpurun_out/r2g
timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r2g/bench_persist.json 2> gpurun_out/r2g/bench_persist.err
MICO_GEMM_VARIANT=4 timeout 600 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r2g/bench_nopersist.json 2> gpurun_out/r2g/bench_nopersist.err
for f in persist nopersist; do python - <<PY
import json
d = json.load(open("gpurun_out/r2g/bench_$f.json"))
print("$f", d["value"], d["ms_per_step"], d["roofline"]["all_gemm"])
for k, v in d["roofline"]["variants"].items(): print("   ", k[:70], round(v["tflops"]), v["launches"])
PY
done
