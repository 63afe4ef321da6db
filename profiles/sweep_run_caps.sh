#!/bin/bash
# A/B of k_invert_band2's hand-over caps (table in xsw_band.hpp): XSW_LONG_RUN_MAX / XSW_LONG_RUN_MAX_CUT = rows of
# band along the a-priori direction beyond which a pixel skips k_invert_band2 and goes straight to the work list (cut: windows
# cut at the last monotone row), XSW_SWEEP_MAX = rows a direction may hold in k_invert_band2's sweep.  Compile-time values: one
# library per configuration is built on the GPU box (XSW_LIB / XSW_EXTRA_FLAGS of xsarsea_amd/_build.py).  Run from the repo root.
S='friendly,anc x1.6,anc x2.5,inc 17-33 anc x1.6,anc x0.3,inc 17-25,anc x0.6,inc 17-33'
mkdir -p build
for cfg in "64 24 64" "128 64 128" "256 256 256"; do
  set -- $cfg
  echo "== LONG_RUN_MAX=$1 LONG_RUN_MAX_CUT=$2 SWEEP_MAX=$3"
  lib=$(pwd)/build/libxsw_caps_$1_$2_$3.so
  XSW_LIB=$lib XSW_EXTRA_FLAGS="-DXSW_LONG_RUN_MAX=$1 -DXSW_LONG_RUN_MAX_CUT=$2 -DXSW_SWEEP_MAX=$3" python3 -m xsarsea_amd._build > /dev/null 2>&1 || exit 1
  XSW_LIB=$lib timeout -k 10 200 python3 profiles/hard_scenes.py --verify --only "$S" 2>/dev/null || exit 1
  XSW_LIB=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('bench', d['value'], 'Mpx/s', d['ms_per_step'], 'ms  band', r['kernel_ms'], r['second_kernel'])"
done
