R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r2p; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2p/stats -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/r2p/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex gemm --output-format csv -d $R/gpurun_out/r2p/fetch -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $R/gpurun_out/r2p/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex gemm --output-format csv -d $R/gpurun_out/r2p/write -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $R/gpurun_out/r2p/write.log 2>&1
cd $R
python tools/pmc_traffic.py gpurun_out/r2p/fetch gpurun_out/r2p/write gpurun_out/r2p/r02_gemm_hbm_traffic.json > /dev/null 2>&1
f=$(find gpurun_out/r2p/stats -name "*kernel_stats.csv" | head -1); python tools/prof_summary.py $f 45 > gpurun_out/r2p/r02_kernel_stats.txt; cp $f gpurun_out/r2p/r02_kernel_stats.csv
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --gemm-detail > gpurun_out/r2p/bench_detail.json 2> gpurun_out/r2p/r02_gemm_detail.txt
# keep only small files
find gpurun_out/r2p -name "*.csv" -size +3M -delete; find gpurun_out/r2p -name "*.db" -delete
head -30 gpurun_out/r2p/r02_kernel_stats.txt; cat gpurun_out/r2p/r02_gemm_hbm_traffic.json | head -40
