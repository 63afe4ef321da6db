#!/bin/bash
# VALU/SALU/VMEM instruction counts of k_invert on the benchmark workload (20000 x 20000), one rocprofv3 --pmc pass.
# Run on the GPU box from the repo root:  bash profiles/collect_sq.sh   -> gpurun_out/sq/summary.json
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/sq
XSW_TRAFFIC_ONLY_INVERT=1 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/sq -- python3 $R/profiles/traffic_driver.py > $R/gpurun_out/sq.log 2>&1
python3 - <<PY
import csv, glob, json, collections
R="$R"
acc=collections.defaultdict(float); n=collections.Counter()
for f in glob.glob(f"{R}/gpurun_out/sq/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_invert" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
px=20000*20000
out={k: v/max(n[k],1) for k,v in acc.items()}
out["launches"]=max(n.values()) if n else 0
out["per_pixel"]={k: out[k]/px for k in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_VMEM_RD","SQ_ACTIVE_INST_VALU") if k in out}
json.dump(out,open(f"{R}/gpurun_out/sq/summary.json","w"),indent=1)
print(json.dumps(out,indent=1))
PY
