#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters, as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE passes (TCC slots), counter unit = KiB, and the read side CALIBRATED on a kernel of known
# byte count with the same access width (k_detrend: 4-B-per-lane loads) because FETCH_SIZE under-reports on gfx950.
# Run on the GPU box from the repo root:  bash profiles/collect_traffic.sh   -> gpurun_out/traffic/*.csv
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/traffic/$c -- python3 $R/profiles/traffic_driver.py > $R/gpurun_out/traffic_$c.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections
R="$R"
vals=collections.defaultdict(dict)
for c in ("FETCH_SIZE","WRITE_SIZE"):
    for f in glob.glob(f"{R}/gpurun_out/traffic/{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            name="k_detrend" if "k_detrend" in k else ("k_invert" if "k_invert" in k else None)  # k_invert_band + k_invert_list of one inversion are summed
            if name and r["Counter_Name"]==c:
                vals[name][c]=vals[name].get(c,0.0)+float(r["Counter_Value"])
                if name=="k_invert":  # and per kernel of the chain (k_invert_band / k_invert_band2 / k_invert_list)
                    sub=k.split("<")[0].split("::")[-1]
                    vals[sub][c]=vals[sub].get(c,0.0)+float(r["Counter_Value"])
n=20000*20000
cal_read = (4.0*n)/(vals["k_detrend"]["FETCH_SIZE"]*1024)      # true bytes / reported bytes, 4-B-per-lane loads
cal_write= (8.0*n)/(vals["k_detrend"]["WRITE_SIZE"]*1024)
inv_r=vals["k_invert"]["FETCH_SIZE"]*1024; inv_w=vals["k_invert"]["WRITE_SIZE"]*1024
out={"raw_KiB":vals,"calibration":{"read_true_over_reported":cal_read,"write_true_over_reported":cal_write,
      "method":"k_detrend 20000x20000 f32->f64: 1.6e9 B read, 3.2e9 B written"},
     "k_invert_20000x20000":{"fetch_bytes_reported":inv_r,"write_bytes_reported":inv_w,
      "fetch_bytes_calibrated":inv_r*cal_read,"write_bytes_calibrated":inv_w*cal_write,
      "hbm_bytes_per_launch":inv_r*cal_read+inv_w*cal_write,"algorithmic_bytes_per_launch":24.0*n}}
out["per_kernel_20000x20000"]={k:{"fetch_bytes_calibrated":v.get("FETCH_SIZE",0.0)*1024*cal_read,"write_bytes_calibrated":v.get("WRITE_SIZE",0.0)*1024*cal_write,
      "hbm_bytes_per_launch":v.get("FETCH_SIZE",0.0)*1024*cal_read+v.get("WRITE_SIZE",0.0)*1024*cal_write} for k,v in vals.items() if k.startswith("k_invert_")}
json.dump(out,open(f"{R}/gpurun_out/traffic/summary.json","w"),indent=1)
print(json.dumps(out,indent=1))
PY
rm -rf $R/gpurun_out/traffic/FETCH_SIZE $R/gpurun_out/traffic/WRITE_SIZE   # raw per-dispatch CSVs: large; the summary stays
