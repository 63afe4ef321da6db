"""Is the statistics instantiation deterministic?  The same tile inverted N times in chunks with xsw_stats_enable(1) (and N times in
production mode), each result compared with ONE production launch of the whole tile.  python3 profiles/debug_stats_mode_repeat.py [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xsarsea_amd import _lib, multi_gpu

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
lines, samples = 2000, 20000
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
_lut, co = bench.build_product_lut(None, "cmod5n")
ctx.upload_luts(co=co)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx.set_stream(stream.cuda_stream)
inc, s_vv, anc = bench.make_scene(lines, samples, 8000, 0, 20260322, dev)
whole = torch.empty((lines, samples), dtype=torch.complex64, device=dev)
ctx.invert_raw(lines, samples, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, inc.data_ptr(), s_vv.data_ptr(), None, None, anc.data_ptr(), whole.data_ptr(), None,
               algo=_lib.ALGO_PRUNED)
torch.cuda.synchronize()
bits = lambda t: torch.view_as_real(t).view(torch.int32)
pipe = multi_gpu.TiledPipeline(lines, samples, dual=False, device=dev, n_chunks=8)
for mode in (1, 0, 1):
    ctx.stats_enable(bool(mode))
    for r in range(reps):
        res = multi_gpu.invert_tiled_device(ctx, inc, s_vv, anc, lines, pipeline=pipe, algo=_lib.ALGO_PRUNED)
        torch.cuda.synchronize()
        d = (bits(whole) != bits(res)).any(dim=-1)
        n = int(d.sum().item())
        rows = d.any(dim=1).nonzero().flatten()
        print(f"stats {mode} rep {r}: differing pixels {n}" + (f" in lines {rows[:4].tolist()}..{rows[-2:].tolist()}" if n else ""), flush=True)
ctx.stats_enable(False)
