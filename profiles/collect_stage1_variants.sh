#!/bin/bash
# Round 5, review item 3: k_invert_band under the counters for variant builds of libxsw.so (build/var/lib_*.so: production, stage 1
# alone, one ray instead of three, one side step, the float32-screening timing experiment) on the benchmark workload (20000 x 20000).
# One rocprofv3 --pmc pass per library (instruction counts + active cycles); per pixel figures and the kernel's share of the GPU time.
#   bash profiles/collect_stage1_variants.sh "<libs>"   -> gpurun_out/stage1_variants.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIBS=${1:-$(ls $R/build/var/lib_*.so)}
OUT=$R/gpurun_out/stage1_variants
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
for lib in $LIBS; do
  tag=$(basename $lib .so)
  XSW_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU_INT32 GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT/$tag -- python3 $R/profiles/traffic_driver.py > $OUT/$tag.log 2>&1 || echo "$tag failed"
  python3 - <<PY
import csv, glob, collections
px = 20000 * 20000
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$OUT/$tag/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_invert_band<" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
g = lambda k: acc[k] / max(n[k], 1)
cyc = g("GRBM_GUI_ACTIVE") / 8.0
print(f"$tag: k_invert_band per pixel: VALU {g('SQ_INSTS_VALU')/px:6.2f}  SALU {g('SQ_INSTS_SALU')/px:6.2f}  VMEM_RD {g('SQ_INSTS_VMEM_RD')/px:5.2f}  LDS {g('SQ_INSTS_LDS')/px:5.2f}  INT32 {g('SQ_INSTS_VALU_INT32')/px:6.2f}"
      f"  | GPU-active {cyc/2.4e6:7.2f} ms at 2.4 GHz  VALU issue {g('SQ_INSTS_VALU')*4/(1024*cyc) if cyc else 0:5.3f}")
PY
  rm -rf $OUT/$tag
done
