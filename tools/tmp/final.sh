mkdir -p gpurun_out/final
timeout 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/final/gpu_tests.log
timeout 420 python bench.py > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench.err
tail -5 gpurun_out/final/gpu_tests.log; head -c 600 gpurun_out/final/bench_line.json
