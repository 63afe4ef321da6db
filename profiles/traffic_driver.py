#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (run under `rocprofv3 --pmc FETCH_SIZE` and, separately,
`--pmc WRITE_SIZE`; see collect_traffic.sh).  One launch each of
  * k_detrend on a 20000x20000 float32 raster -> float64   (CALIBRATION: known 1.6 GB read, 3.2 GB written,
    same 4-B-per-lane loads / 8-B-per-lane stores as the inversion's raster accesses), and
  * k_invert (pruned) on the benchmark scene 20000x20000.
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from xsarsea_amd import _lib  # noqa: E402

lines = samples = int(os.environ.get("XSW_TRAFFIC_N", "20000"))
dev = torch.device("cuda", 0)
lut, co = bench.build_product_lut()
ctx = _lib.Context(0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx.set_stream(stream.cuda_stream)
ctx.upload_luts(co=co)
inc, s_vv, anc = bench.make_scene(lines, samples, lines, 0, 20260322, dev)
out = torch.empty((lines, samples), dtype=torch.complex64, device=dev)
det = torch.empty((lines, samples), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
ctx.detrend_raw(lines, samples, _lib.XSW_F32, _lib.XSW_F64, _lib.MEM_DEVICE, s_vv.data_ptr(), np.ones(samples), det.data_ptr())
if os.environ.get("XSW_TRAFFIC_DUAL"):  # the dual-pol workload (bench config 3): cross-pol LUT, sigma0_vh + per-pixel dsig, fused select
    from xsarsea_amd.windspeed import _engine, get_model
    ctx.upload_luts(cr=_engine._cr_dict(get_model("gmf_s1_v2")._lut(units="dB")))
    s_vh, dsig = bench.make_crosspol(inc, anc, 777, dev)
    out_dual = torch.empty((lines, samples), dtype=torch.complex64, device=dev)
    torch.cuda.synchronize()
    ctx.invert_raw(lines, samples, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, inc.data_ptr(), s_vv.data_ptr(), s_vh.data_ptr(), dsig.data_ptr(),
                   anc.data_ptr(), out.data_ptr(), out_dual.data_ptr(), algo=_lib.ALGO_PRUNED, dual_select=True)
else:
    ctx.invert_raw(lines, samples, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, inc.data_ptr(), s_vv.data_ptr(), None, None,
                   anc.data_ptr(), out.data_ptr(), None, algo=_lib.ALGO_PRUNED)
torch.cuda.synchronize()
print("done", lines, samples)
