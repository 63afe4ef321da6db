#!/bin/bash
# Issue / stall / texture-path counters of k_invert on the benchmark workload (20000 x 20000), several rocprofv3 --pmc passes
# (kernel-trace/stats are NOT combined with --pmc).  Run on the GPU box from the repo root:
#   bash profiles/collect_counters.sh [tag]      -> gpurun_out/counters_<tag>/summary.json
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}
OUT=$R/gpurun_out/counters_$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
i=0
for set in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
 "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU" \
 "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
 "SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVES SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/profiles/traffic_driver.py > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, json, collections
px=20000*20000
res={}
for kern in ("k_invert_band<", "k_invert_band2<", "k_invert_blocks<", "k_invert_list<", "k_invert<"):
    acc=collections.defaultdict(float); n=collections.Counter()
    for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    if not acc: continue
    out={k: v/max(n[k],1) for k,v in acc.items()}
    out["per_pixel"]={k: round(v/px,3) for k,v in out.items() if k.startswith(("SQ_INSTS","SQ_ACTIVE","SQ_WAIT","SQ_WAVE_CYCLES","TA_","TCP_","SQ_INST_CYCLES","SQ_THREAD"))}
    res[kern.rstrip("<")]=out
json.dump(res,open("$OUT/summary.json","w"),indent=1)
print(json.dumps({k: v["per_pixel"] | {"GRBM_GUI_ACTIVE": v.get("GRBM_GUI_ACTIVE")} for k,v in res.items()},indent=1))
PY
rm -rf $OUT/p[0-9]*   # the raw per-dispatch CSVs are large (gpurun_out travels back: 64 MiB cap); the summary stays
