R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_clk; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in shipped pcab1 pcab4; do
  if [ $v = shipped ]; then L=$R/mico_amd/libmico_hip.so; else L=$R/tools/probes/bin/libmico_$v.so; fi
  MICO_HIP_LIB=$L rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex gemm_pc --output-format csv -d $O/$v -- python $R/tools/gemm_bench.py --dtype fp16 --only dw --iters 10 > $O/$v.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections
for v in ("shipped", "pcab1", "pcab4"):
    cc = glob.glob(f"gpurun_out/pmc_clk/{v}/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(f"gpurun_out/pmc_clk/{v}/**/*kernel_trace.csv", recursive=True)
    cyc = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        cyc[r["Dispatch_Id"]] = (float(r["Counter_Value"]), r["Grid_Size"])
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    byg = collections.defaultdict(list)
    for d, (c, g) in cyc.items():
        if d in dur: byg[g].append(c / dur[d])
    print(v, {g: round(sum(x) / len(x), 3) for g, x in byg.items()}, "GHz (GRBM_GUI_ACTIVE / ns), n =", {g: len(x) for g, x in byg.items()})
PY
