"""Hard-scene table: the search kernels' times on 4000 x 20000-pixel bands of the bench generator with the a-priori
wind scaled and / or near-range incidences.  Run on the GPU box from the repo root:

    python3 profiles/hard_scenes.py [--verify] > gpurun_out/hard_scenes.txt

One line per scene: per-kernel ms (events inside the library: xsw_timing_read), the share of the pixels handed to
k_invert_band2 / left to k_invert_list, Mpixels/s, scored candidates per pixel (statistics instantiation), and -- with
--verify -- the number of output values that differ from the LDS-tiled exhaustive sweep (an independent kernel; must be 0).
A/B switches of the library are read from the environment as usual (XSW_LONG_RUN, XSW_NO_TAIL_CUT, ...)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (make_scene, build_product_lut)
from xsarsea_amd import _lib  # noqa: E402

SCENES = (("friendly", (30.0, 46.0), 1.0, 0.0), ("outliers 1%", (30.0, 46.0), 1.0, 0.01), ("outliers 5%", (30.0, 46.0), 1.0, 0.05),
          ("anc x0.6", (30.0, 46.0), 0.6, 0.0), ("anc x0.3", (30.0, 46.0), 0.3, 0.0),
          ("anc x1.3", (30.0, 46.0), 1.3, 0.0), ("anc x1.6", (30.0, 46.0), 1.6, 0.0), ("anc x2.5", (30.0, 46.0), 2.5, 0.0),
          ("inc 17-33", (17.0, 33.0), 1.0, 0.0), ("inc 17-33 anc x0.6", (17.0, 33.0), 0.6, 0.0), ("inc 17-33 anc x1.6", (17.0, 33.0), 1.6, 0.0),
          ("inc 17-25", (17.0, 25.0), 1.0, 0.0), ("inc 17-33 outliers 5%", (17.0, 33.0), 1.0, 0.05))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", type=int, default=4000)
    ap.add_argument("--samples", type=int, default=20000)
    ap.add_argument("--verify", action="store_true", help="compare every scene with the exhaustive sweep")
    ap.add_argument("--only", default=None, help="comma-separated scene names")
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    _lut, co = bench.build_product_lut(None, "cmod5n")
    ctx.upload_luts(co=co)
    lines, samples = args.lines, args.samples
    o = torch.empty((lines, samples), dtype=torch.complex64, device=device)
    o2 = torch.empty((lines, samples), dtype=torch.complex64, device=device) if args.verify else None
    only = set(args.only.split(",")) if args.only else None
    for name, inc_range, scale, outl in SCENES:
        if only and name not in only:
            continue
        inc, s_vv, anc = bench.make_scene(lines, samples, 20000, 8000, 20260320 + 7, device, inc_range=inc_range, anc_scale=scale, outlier_frac=outl)
        torch.cuda.synchronize()

        def run(out, algo=_lib.ALGO_PRUNED):
            ctx.invert_raw(lines, samples, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, inc.data_ptr(), s_vv.data_ptr(), None, None,
                           anc.data_ptr(), out.data_ptr(), None, algo=algo)
        run(o)
        ctx.synchronize()
        ctx.timing_enable(True)
        run(o)
        run(o)
        tm = ctx.timing()
        ctx.timing_enable(False)
        ctx.stats_enable(True)
        run(o)
        st = ctx.stats()
        ctx.stats_enable(2)  # the production chain with per-kernel counters: candidates k_invert_band2 / k_invert_blocks score per pixel handed
        run(o)
        ch = ctx.stats_chain()
        ctx.stats_enable(False)
        n = max(tm["launches"], 1)
        b, b2, bl, ls = tm["first_kernel_ms"] / n, tm["band2_kernel_ms"] / n, tm["blocks_kernel_ms"] / n, tm["second_kernel_ms"] / n
        px = lines * samples
        line = (f"{name:<22s} band {b:8.2f} ms  band2 {b2:8.2f} ms ({100 * tm['last_band2_pixels'] / px:6.2f} %)"
                f"  blocks {bl:8.2f} ms ({100 * tm['last_blocks_pixels'] / px:6.2f} %)  list {ls:8.2f} ms"
                f"  -> {px / (b + b2 + bl + ls) / 1e3:8.0f} Mpx/s  listed {100 * tm['last_list_pixels'] / px:6.2f} %"
                f"  cand/px {st['cand_co'] / max(st['pixels_co'], 1):8.1f}  exact {st.get('pixels_exact', 0)}"
                f"  chain cand/px b2 {ch['cand_band2'] / max(tm['last_band2_pixels'], 1):7.1f} blk {ch['cand_blocks'] / max(tm['last_blocks_pixels'], 1):7.1f}"
                f" refined {100.0 * ch['pixels_refined'] / max(tm['last_band2_pixels'], 1):5.1f} %")
        if args.verify:
            run(o)
            run(o2, _lib.ALGO_EXHAUSTIVE)
            ctx.synchronize()
            torch.cuda.synchronize()
            diff = int((torch.view_as_real(o).view(torch.int32) != torch.view_as_real(o2).view(torch.int32)).sum().item())
            line += f"  differing values vs exhaustive {diff}"
        print(line, flush=True)
        del inc, s_vv, anc


if __name__ == "__main__":
    main()
