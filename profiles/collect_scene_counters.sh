#!/bin/bash
# Issue / stall counters of every kernel of the chain on ONE hard scene (profiles/scene_driver.py), rocprofv3 --pmc passes.
#   bash profiles/collect_scene_counters.sh "anc x2.5" tag   -> gpurun_out/scene_<tag>.json (per kernel, per pixel it handled)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export XSW_SCENE="${1:-anc x2.5}"
TAG=${2:-x}
OUT=$R/gpurun_out/scene_$TAG
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
i=0
for set in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
 "SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVES SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_VALU_INT32" \
 "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/profiles/scene_driver.py > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, json, collections
px=[int(x) for x in open("$OUT/p1.log").read().split("PIXELS")[1].split()[:4]]
handled={"k_invert_band<": px[0], "k_invert_band2<": px[1], "k_invert_blocks<": px[2], "k_invert_list<": px[3]}
res={"scene": "$XSW_SCENE", "pixels": handled}
for kern, n_px in handled.items():
    acc=collections.defaultdict(float); n=collections.Counter()
    for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    if not acc or not n_px: continue
    out={k: v/max(n[k],1) for k,v in acc.items()}
    cyc=out.get("GRBM_GUI_ACTIVE",0)/8.0
    d={"per_pixel": {k: round(v/n_px,2) for k,v in out.items() if k.startswith(("SQ_INSTS","SQ_WAVE_CYCLES","SQ_WAIT_ANY","TCP_","TCC_","SQ_INST_LEVEL_VMEM","SQ_INST_CYCLES_VMEM"))}}
    if cyc and out.get("TA_TA_BUSY_sum"): d["ta_busy_frac"]=round(out["TA_TA_BUSY_sum"]/(256*cyc),3)
    if out.get("SQ_INST_LEVEL_VMEM") and out.get("SQ_INSTS_VMEM_RD"): d["avg_vmem_latency_cycles"]=round(out["SQ_INST_LEVEL_VMEM"]/out["SQ_INSTS_VMEM_RD"],0)
    if cyc:
        d["valu_issue_frac"]=round(out.get("SQ_INSTS_VALU",0)*4/(1024*cyc),3)
        d["gpu_cycles"]=cyc
    if out.get("SQ_WAVE_CYCLES"): d["wait_frac"]=round(out.get("SQ_WAIT_ANY",0)/out["SQ_WAVE_CYCLES"],3)
    if out.get("SQ_WAVES"): d["waves"]=out["SQ_WAVES"]
    if out.get("SQ_LEVEL_WAVES") and cyc: d["avg_waves_in_flight"]=round(out["SQ_LEVEL_WAVES"]/cyc,1)
    res[kern.rstrip("<")]=d
json.dump(res,open("$OUT.json","w"),indent=1)
print(json.dumps(res,indent=1))
PY
rm -rf $OUT
