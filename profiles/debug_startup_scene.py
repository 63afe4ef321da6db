"""bench.py's start-up (context, device-built LUT, host-built LUT upload) followed at once by the scene generation, as N processes do side
by side at the start of an N-rank job: is the first generated tile reproduced by the later ones?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import xsarsea_amd
from xsarsea_amd import _lib
from xsarsea_amd.windspeed import _engine, get_model
dev = torch.device("cuda", 0)
lut, co = bench.build_product_lut(None, "cmod5n")
ctx = _lib.Context(0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx.set_stream(stream.cuda_stream)
plan = get_model("gmf_cmod5n").device_lut_plan()
_engine.DeviceLut("gmf_cmod5n", plan[0], plan[1], plan[2], key=None).build(ctx)
ctx.synchronize()
ctx.upload_luts(co=co)
ctx.synchronize()
first = bench.make_scene(2000, 20000, 8000, 0, 20260322, dev)   # no synchronisation behind it, as bench.py had it
out = torch.empty((2000, 20000), dtype=torch.complex64, device=dev)
codes = torch.empty((8000, 20000), dtype=torch.int32, device=dev)
full = torch.empty((8000, 20000), dtype=torch.complex64, device=dev)
hello = torch.zeros((4, 20000), dtype=torch.complex64, device=dev)
ctx.invert_raw(250, 20000, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, first[0].data_ptr(), first[1].data_ptr(), None, None, first[2].data_ptr(), out.data_ptr(), None,
               algo=_lib.ALGO_PRUNED)
torch.cuda.synchronize()
for r in range(3):
    again = bench.make_scene(2000, 20000, 8000, 0, 20260322, dev)
    torch.cuda.synchronize()
    for name, a, b in zip(("inc", "s_vv", "anc"), first, again):
        ai = (torch.view_as_real(a) if a.is_complex() else a).contiguous().view(torch.int32)
        bi = (torch.view_as_real(b) if b.is_complex() else b).contiguous().view(torch.int32)
        d = ai != bi
        d = d.any(dim=-1) if d.dim() == 3 else d
        n = int(d.sum().item())
        if n:
            rows = d.any(dim=1).nonzero().flatten()
            print(f"rep {r}: {name}: {n} values differ from the first generation, lines {rows[:3].tolist()}..{rows[-2:].tolist()}", flush=True)
print("done", flush=True)
