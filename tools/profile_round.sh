#!/bin/bash
# Collect the per-round evidence that goes into profiles/ (run on the GPU box through gpurun):
#   tools/profile_round.sh r02        -> gpurun_out/prof_r02/{r02_kernel_stats.txt,.csv, r02_gemm_hbm_traffic.json,
#                                         r02_gemm_detail.txt, r02_sq_counters.txt, r02_bench_line.json}
# Kernel trace + stats and the two PMC passes are separate rocprofv3 runs (counters never share a run with a trace).
set -u
TAG=${1:-rXX}; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/prof_$TAG
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-extras --no-comm"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 3 --warmup 2 > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex gemm --output-format csv -d $O/fetch -- $B --steps 1 --warmup 0 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex gemm --output-format csv -d $O/write -- $B --steps 1 --warmup 0 > $O/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
    --kernel-include-regex "gemm|attn" --output-format csv -d $O/sq -- $B --steps 1 --warmup 0 > $O/sq.log 2>&1
cd $R
python tools/pmc_traffic.py $O/fetch $O/write $O/${TAG}_gemm_hbm_traffic.json > /dev/null 2>&1
python tools/pmc_sq.py $O/sq $O/${TAG}_sq_counters.txt > /dev/null 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
python tools/prof_summary.py $f 45 > $O/${TAG}_kernel_stats.txt; cp $f $O/${TAG}_kernel_stats.csv
$B --steps 10 --warmup 3 --gemm-detail > $O/bench_detail.json 2> $O/${TAG}_gemm_detail.txt
MICO_BENCH_FULL=$O/${TAG}_bench_full.json python bench.py > $O/${TAG}_bench_line.json 2> $O/bench_line.err
python tools/probes/sync_probe.py > $O/${TAG}_sync_probe.txt 2>&1
# the round's A/B: round 5's step (pair stash, direct backward) against the default (pre-activation stash where memory limits, staged backward), alternating
bash tools/probes/stash_ab.sh $O > $O/${TAG}_stash_staged_ab.txt 2>&1
if [ "${PROFILE_VIDCAP:-0}" = 1 ]; then
  # BASELINE configs[4] shapes: the fp8 mode and the same step in bf16 on the same box (two alternating pairs), + the fp8 step's kernel table
  for i in 1 2; do
    python bench.py --workload vid_cap_fp8 --no-cpu-baseline --no-comm > $O/${TAG}_bench_line_fp8_vidcap$i.json 2> $O/bench_fp8.err
    python bench.py --workload vid_cap_fp8 --dtype bf16 --no-cpu-baseline --no-comm > $O/${TAG}_bench_line_bf16_vidcap$i.json 2> $O/bench_bf16.err
  done
  mv $O/${TAG}_bench_line_fp8_vidcap2.json $O/${TAG}_bench_line_fp8_vidcap.json; mv $O/${TAG}_bench_line_bf16_vidcap2.json $O/${TAG}_bench_line_bf16_vidcap.json
  (cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fp8 -- python $R/bench.py --workload vid_cap_fp8 --no-cpu-baseline --no-extras --no-comm --steps 3 --warmup 2 > $O/stats_fp8.log 2>&1)
  f8=$(find $O/stats_fp8 -name "*kernel_stats.csv" | head -1)
  python tools/prof_summary.py $f8 30 > $O/${TAG}_kernel_stats_fp8_vidcap.txt
fi
if [ "${PROFILE_CALIB:-0}" = 1 ]; then
  # the vendor library (hipBLASLt behind torch.matmul) on the towers' layer shapes next to mico_gemm, same process, same box
  python tools/probes/hipblaslt_ref.py --m 184269 > $O/${TAG}_hipblaslt_calibration.txt 2>&1
fi
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete
head -30 $O/${TAG}_kernel_stats.txt; cat $O/${TAG}_bench_line.json
