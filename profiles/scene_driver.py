#!/usr/bin/env python3
"""Workload for counter passes on ONE hard scene of profiles/hard_scenes.py (XSW_SCENE="anc x2.5" ...; 4000 x 20000 pixels):
a warm-up launch and one measured launch of the inversion chain.  Prints the pixels each kernel of the chain handled."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "profiles"))
import bench  # noqa: E402
import hard_scenes  # noqa: E402
from xsarsea_amd import _lib  # noqa: E402

name = os.environ.get("XSW_SCENE", "anc x2.5")
sc = [s for s in hard_scenes.SCENES if s[0] == name][0]
lines, samples = int(os.environ.get("XSW_SCENE_LINES", "4000")), 20000
dev = torch.device("cuda", 0)
_lut, co = bench.build_product_lut()
ctx = _lib.Context(0)
ctx.upload_luts(co=co)
inc, s_vv, anc = bench.make_scene(lines, samples, 20000, 8000, 20260320 + 7, dev, inc_range=sc[1], anc_scale=sc[2], outlier_frac=sc[3])
out = torch.empty((lines, samples), dtype=torch.complex64, device=dev)
torch.cuda.synchronize()
ctx.timing_enable(True)
for _ in range(2):
    ctx.invert_raw(lines, samples, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, inc.data_ptr(), s_vv.data_ptr(), None, None,
                   anc.data_ptr(), out.data_ptr(), None, algo=_lib.ALGO_PRUNED)
tm = ctx.timing()
print("PIXELS", lines * samples, tm["last_band2_pixels"], tm["last_blocks_pixels"], tm["last_list_pixels"])
