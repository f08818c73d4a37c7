cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-include-regex smallq --output-format csv -d /tmp/sqc -- python $GRAFT_REPO_ROOT/tools/probes/drop_cost.py > /tmp/sqc.log 2>&1
python - <<'P'
import csv,glob
acc={}
for f in glob.glob('/tmp/sqc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]]=acc.get(r["Counter_Name"],0)+float(r["Counter_Value"])
print(acc, acc.get("SQ_LDS_BANK_CONFLICT",0)/max(acc.get("SQ_LDS_IDX_ACTIVE",1),1))
P
