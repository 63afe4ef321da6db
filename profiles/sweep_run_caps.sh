#!/bin/bash
# A/B of k_invert_band2's run caps: XSW_RUN_MAX / XSW_RUN_MAX_CUT = rows of band along the a-priori direction beyond
# which a pixel skips k_invert_band2 and goes straight to the work list (cut: windows cut at the last monotone row),
# XSW_SWEEP_MAX = rows a direction may hold in k_invert_band2's sweep.  Run on the GPU box from the repo root.
S='friendly,anc x1.6,anc x2.5,inc 17-33 anc x1.6,anc x0.3,inc 17-25,anc x0.6,inc 17-33'
for cfg in "64 24 64" "128 64 128" "256 256 256"; do
  set -- $cfg
  echo "== RUN_MAX=$1 RUN_MAX_CUT=$2 SWEEP_MAX=$3"
  XSW_RUN_MAX=$1 XSW_RUN_MAX_CUT=$2 XSW_SWEEP_MAX=$3 timeout -k 10 200 python3 profiles/hard_scenes.py --verify --only "$S" 2>/dev/null || exit 1
  XSW_RUN_MAX=$1 XSW_RUN_MAX_CUT=$2 XSW_SWEEP_MAX=$3 timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('bench', d['value'], 'Mpx/s', d['ms_per_step'], 'ms  band', r['kernel_ms'], r['second_kernel'])"
done
