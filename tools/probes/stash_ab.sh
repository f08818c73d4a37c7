mkdir -p gpurun_out/r6g
for n in new1 old1 new2 old2 new3 old3; do case $n in old*) E="MICO_MLP_STASH=pair"; A="--direct-backward";; *) E="X=1"; A="";; esac
env $E python bench.py --no-extras --no-comm --no-cpu-baseline --steps 10 --warmup 2 $A > gpurun_out/r6g/$n.json 2> gpurun_out/r6g/$n.err
python - <<PY
import json,re
s=open("gpurun_out/r6g/$n.err").read()
d=json.loads(re.search(r"BENCH_FULL_JSON (.*)", s).group(1))
r=d["roofline"]; ms=d["ms_per_step"]
g=r["all_gemm"]["share_of_step_time"]*ms
print("$n", round(d["value"],2), round(ms,1), "gemm", round(g,1), "other", round(ms-g,1), "plan", d["tower_plan"]["mlp_blocks_kept"])
for k,v in list(r["variants"].items())[:4]:
    print("    ", k[:40], k[-22:], v["launches"], round(v["avg_ms"],3), round(v["tflops"],1))
PY
done
