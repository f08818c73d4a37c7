#!/bin/bash
# A/B on one box: functional.DkvSession (one dK/dV buffer for the own set the captioning pass and the ITM triplet share) against a buffer per pass
# summed by autograd (MICO_DKV_PER_PASS=1), and the soft memory budget one / two blocks higher with the session on.
#   gpurun -- 'bash tools/probes/dkv_ab.sh'   -> gpurun_out/dkv/*.json
mkdir -p gpurun_out/dkv
B="python bench.py --no-cpu-baseline --no-extras --no-comm --steps 5 --warmup 3"
MICO_DKV_PER_PASS=1 timeout 400 $B > gpurun_out/dkv/perpass3.json 2> gpurun_out/dkv/perpass3.err
timeout 400 $B > gpurun_out/dkv/inplace3.json 2> gpurun_out/dkv/inplace3.err
MICO_HBM_SOFT_FRAC=0.88 timeout 400 $B > gpurun_out/dkv/inplace_088.json 2> gpurun_out/dkv/inplace_088.err
MICO_HBM_SOFT_FRAC=0.89 timeout 400 $B > gpurun_out/dkv/inplace_089.json 2> gpurun_out/dkv/inplace_089.err
MICO_DKV_PER_PASS=1 timeout 400 $B > gpurun_out/dkv/perpass4.json 2> gpurun_out/dkv/perpass4.err
for f in gpurun_out/dkv/perpass3 gpurun_out/dkv/inplace3 gpurun_out/dkv/inplace_088 gpurun_out/dkv/inplace_089 gpurun_out/dkv/perpass4; do python - $f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], round(d['value'],2), round(d['ms_per_step'],1), round(d['peak_mem_gb'],2), round(d['peak_reserved_gb'],2), d['tower_plan']['mlp_blocks_kept'], d['allocator'], round(d['roofline']['all_gemm']['share_of_step_time'],4))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
