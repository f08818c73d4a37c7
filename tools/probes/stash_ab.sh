#!/bin/bash
# Same-box A/B of round 6's step schedule (run on the GPU box): round 5's form - every kept block keeps gelu + gelu' (MICO_MLP_STASH=pair), one
# backward over every graph of the step (--direct-backward) - against the default - the fc1 pre-activation alone where memory limits the kept
# blocks, BERT passes differentiated inside the forward (staged) - alternating, 10 timed steps each.  Prints samples/s, ms per step, the time inside
# timed GEMM launches and outside them, the kept blocks and the four largest GEMM classes.
O=${1:-gpurun_out/r6g}; mkdir -p $O
for n in new1 old1 new2 old2 new3 old3; do case $n in old*) E="MICO_MLP_STASH=pair"; A="--direct-backward";; *) E="X=1"; A="";; esac
env $E python bench.py --no-extras --no-comm --no-cpu-baseline --steps 10 --warmup 2 $A > $O/ab_$n.json 2> $O/ab_$n.err
python - <<PY
import json,re
s=open("$O/ab_$n.err").read()
d=json.loads(re.search(r"BENCH_FULL_JSON (.*)", s).group(1))
r=d["roofline"]; ms=d["ms_per_step"]
g=r["all_gemm"]["share_of_step_time"]*ms
print("$n", round(d["value"],2), "samples/s", round(ms,1), "ms/step; in GEMM launches", round(g,1), "outside", round(ms-g,1), "; kept MLP blocks", d["tower_plan"]["mlp_blocks_kept"], d["tower_plan"]["mlp_stash"], "; peak GiB", round(d["peak_mem_gb"],1), "/", round(d["peak_reserved_gb"],1))
for k,v in list(r["variants"].items())[:4]:
    print("    ", k[:34], k[-22:], v["launches"], "launches", round(v["avg_ms"],3), "ms", round(v["tflops"],1), "TFLOP/s")
PY
done
