mkdir -p gpurun_out/r05z
for lr in 4 5 6; do XSW_LONG_RUN=$lr python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/r05z/bench_lr$lr.json 2> gpurun_out/r05z/bench_lr.err && python - <<PY
import json
d=json.loads(open("gpurun_out/r05z/bench_lr$lr.json").read().strip().splitlines()[-1]); r=d["roofline"]
print($lr, d["value"], d["ms_per_step"], r["kernel_ms"], r["second_kernel"]["k_invert_band2_ms"], r["second_kernel"]["pixels_to_band2_last_launch"])
PY
done
for lr in 5 4; do XSW_LONG_RUN=$lr python bench.py --config 3 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r05z/bench_dual_lr$lr.json 2>> gpurun_out/r05z/bench_lr.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r05z/bench_dual_lr$lr.json").read().strip().splitlines()[-1]); print("dual", $lr, d["value"], d["ms_per_step"])
PY
done
