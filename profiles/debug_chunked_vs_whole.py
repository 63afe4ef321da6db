"""Chunked coded inversion (multi_gpu.TiledPipeline, world 1) vs ONE launch of the same tile: every pixel must be bit-equal.
   python3 profiles/debug_chunked_vs_whole.py [tile_lines] [n_chunks] [stats 0/1]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xsarsea_amd import _lib, multi_gpu

lines = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n_chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
stats = len(sys.argv) > 3 and sys.argv[3] == "1"
samples = 20000
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
if stats:
    ctx.stats_enable(True)
pipe = multi_gpu.TiledPipeline(lines, samples, dual=False, device=dev, n_chunks=n_chunks)
res = multi_gpu.invert_tiled_device(ctx, inc, s_vv, anc, lines, pipeline=pipe, algo=_lib.ALGO_PRUNED)
torch.cuda.synchronize()
full = res[0] if isinstance(res, tuple) else res
bits = lambda t: torch.view_as_real(t).view(torch.int32)
d = (bits(whole) != bits(full)).any(dim=-1)
n = int(d.sum().item())
print(f"tile {lines} lines, {n_chunks} chunks, stats {int(stats)}: differing pixels {n}")
if n:
    idx = d.nonzero()[:12].tolist()
    for l, s in idx:
        print("  line", l, "sample", s, "whole", complex(whole[l, s].item()), "chunked", complex(full[l, s].item()), "chunk", l * n_chunks // lines)
    rows = d.any(dim=1).nonzero().flatten()
    print("  lines with differences:", rows[:20].tolist(), "... count", int(rows.numel()))
