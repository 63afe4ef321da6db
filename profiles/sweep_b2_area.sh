#!/bin/bash
# A/B of the hand-over threshold k_invert_band2 -> k_invert_blocks (XSW_B2_AREA: band candidates = run x directions) on the hard scenes
for a in 1000000 8192 4096 2048 1024 512 256; do echo "== XSW_B2_AREA=$a"; XSW_B2_AREA=$a python3 profiles/hard_scenes.py --only "friendly,outliers 5%,anc x0.3,anc x2.5,inc 17-33 anc x1.6,anc x0.6,anc x1.6" 2>&1 | grep -v amdgpu.ids; done
