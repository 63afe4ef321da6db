#!/bin/bash
# Round-2 evidence bundle, run on the GPU box from the repo root:  bash profiles/collect_r02.sh
#   1. rocprofv3 --kernel-trace --stats of the default bench command (per-kernel average durations)
#   2. issue / stall / texture-path counters of the two inversion kernels (collect_counters.sh; --pmc passes only)
#   3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes, read side calibrated on k_detrend (collect_traffic.sh)
# Everything lands in gpurun_out/r02/ ; copy what is to be judged into profiles/ (tracked).
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r02
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r02/kt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02/kt -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/r02/bench_under_rocprof.json 2> $R/gpurun_out/r02/bench_under_rocprof.err
cp $R/gpurun_out/r02/kt/*/*kernel_stats.csv $R/gpurun_out/r02/kernel_stats.csv
grep -E "Kind|k_invert|k_detrend|k_nesz|k_pad|k_gmf|k_lut|k_to_db|k_mono|k_transpose" $R/gpurun_out/r02/kt/*/*kernel_trace.csv | head -60 > $R/gpurun_out/r02/kernel_trace_xsw.csv
rm -rf $R/gpurun_out/r02/kt
cd $R
bash profiles/collect_counters.sh r02 > gpurun_out/r02/counters.log 2>&1
cp gpurun_out/counters_r02/summary.json gpurun_out/r02/pmc_counters_summary.json
bash profiles/collect_traffic.sh > gpurun_out/r02/traffic.log 2>&1
cp gpurun_out/traffic/summary.json gpurun_out/r02/hbm_traffic_summary.json
ls -la gpurun_out/r02
