#!/bin/bash
# Round-5 evidence bundle, run on the GPU box from the repo root:  XSW_COMMIT=<git rev> bash profiles/collect_r05.sh
#   0. provenance: SHA-256 of the device code (.hip_fatbin) of the libxsw.so everything below runs on + the commit
#   1. default bench line + the other BASELINE workloads on one GPU
#   2. rocprofv3 --kernel-trace --stats of the default bench command (per-kernel average durations)
#   3. issue / stall / texture-path counters of the inversion kernels (collect_counters.sh; --pmc passes only)
#   4. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes, read side calibrated on k_detrend (collect_traffic.sh)
#   5. per-kernel register / LDS / scratch table from the code object
# Every counter summary is stamped with `measured_on` = {code_sha256, commit}; bench.py reports counter-derived fields only
# when that hash equals the hash of the library it has loaded (else null + a stale_profile note).
# Everything lands in gpurun_out/r05/ ; copy what is to be judged into profiles/ (tracked).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
SHA=$(python3 -m xsarsea_amd._build --code-sha 2>/dev/null | tail -1)
echo "{\"code_sha256\": \"$SHA\", \"commit\": \"${XSW_COMMIT:-unknown}\"}" > $O/measured_on.json
cat $O/measured_on.json
python3 -m xsarsea_amd._build --resources $O/kernel_resources.tsv > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $O/kt/*/*kernel_stats.csv $O/kernel_stats.csv
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("$O/kt/*/*kernel_trace.csv")[0])))
by = collections.defaultdict(list)
for r in rows:
    k = r["Kernel_Name"]
    if "k_invert" in k or "k_expand" in k:
        by[k.split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open("$O/kernel_trace_durations.txt", "w") as f:
    f.write("# per-dispatch durations (ms, launch order) of the inversion kernels, rocprofv3 --kernel-trace of bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras\n")
    for k, v in by.items():
        f.write(f"{k}: n={len(v)} avg={sum(v)/len(v):.3f} ms  [{', '.join(f'{x:.2f}' for x in v)}]\n")
print(open("$O/kernel_trace_durations.txt").read())
PY
rm -rf $O/kt
cd $R
bash profiles/collect_counters.sh r05 > $O/counters.log 2>&1
bash profiles/collect_traffic.sh > $O/traffic.log 2>&1
python3 - <<PY
import json
on = json.load(open("$O/measured_on.json"))
for src, dst in (("$R/gpurun_out/counters_r05/summary.json", "$O/pmc_counters_summary.json"), ("$R/gpurun_out/traffic/summary.json", "$O/hbm_traffic_summary.json")):
    try:
        d = json.load(open(src))
        d["measured_on"] = on
        json.dump(d, open(dst, "w"), indent=1)
        print("stamped", dst)
    except Exception as e:
        print("missing", src, e)
PY
ls -la $O
