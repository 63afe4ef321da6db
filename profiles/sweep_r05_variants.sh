#!/bin/bash
# Round 5 A/B: variant builds of libxsw.so (build/var/lib_*.so: k_band2_prep occupancy / live-arc unroll, ...) x routing switches, on the
# hard scenes.  Run on the GPU box from the repo root:  bash profiles/sweep_r05_variants.sh "<libs>" "<env settings, ';'-separated>"
LIBS=${1:-$(ls build/var/lib_*.so)}
IFS=';' read -ra ENVS <<< "${2:-XSW_NOP=1}"
SCENES=${SCENES:-"friendly,outliers 5%,anc x0.6,anc x0.3,anc x1.6,anc x2.5,inc 17-33 anc x1.6"}
for lib in $LIBS; do
  for e in "${ENVS[@]}"; do
    echo "== $(basename $lib) [$e]"
    env XSW_LIB=$PWD/$lib $e python3 profiles/hard_scenes.py ${VERIFY:+--verify} --only "$SCENES" 2>&1 | grep -v amdgpu.ids
  done
done
