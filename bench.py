#!/usr/bin/env python3
"""Benchmark of the wind-inversion hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--lines L] [--samples S] [--algo pruned|exhaustive]

One "step" = one pass of `xsw_invert` (CMOD5.N mono-VV, default 0.1 m/s x 1 deg x 0.1 deg LUT) over a
synthetic float32 raster that is already resident in HBM (sigma0 + incidence + complex64 ancillary
wind in, complex64 wind out).  N = 1 workload: the 20000 x 20000 raster BASELINE.json's metric is
quoted on.  N > 1 (launched by torch.distributed.run, one rank per GPU): the raster is row-tiled,
every rank inverts a 20000-line tile of a (20000*N) x 20000 raster (weak scaling) and the output
tiles are gathered on rank 0 over RCCL inside the timed step.

Prints ONE JSON line (rank 0) with the throughput, the HBM roofline of the dominant kernel, and a CPU
baseline (the oracle's C restatement of the reference kernel, timed on this box's cores on a
bounded crop of the same raster).
"""
import argparse
import json
import math
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: required for multi-process RCCL on this host driver

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_READ_PX = 16     # inc f32 + sigma0 f32 + ancillary complex64   
BYTES_WRITE_PX = 8     # complex64 wind


# ------------------------------------------------------------------------------------------ synthetic scene
_C5N = [0.0, -0.6878, -0.7957, 0.338, -0.1728, 0.0, 0.004, 0.1103, 0.0159, 6.7329, 2.7713, -2.2885, 0.4971, -0.725,
        0.045, 0.0066, 0.3222, 0.012, 22.7, 2.0813, 3.0, 8.3659, -3.3428, 1.3236, 6.2437, 2.3893, 0.3249, 4.159, 1.693]


def cmod5n_torch(inc, v, phi_deg):
    """CMOD5.N forward model with torch ops (scene synthesis only; float64)."""
    c = _C5N
    cosphi = torch.cos(torch.deg2rad(phi_deg))
    x = (inc - 40.0) / 25.0
    x2 = x * x
    y0, pn = c[19], c[20]
    a = y0 - (y0 - 1.0) / pn
    b = 1.0 / (pn * (y0 - 1.0) ** (pn - 1.0))
    a0 = c[1] + c[2] * x + c[3] * x2 + c[4] * x * x2
    a1 = c[5] + c[6] * x
    a2 = c[7] + c[8] * x
    gam = c[9] + c[10] * x + c[11] * x2
    s0 = c[12] + c[13] * x
    s = a2 * v
    sig_s0 = torch.sigmoid(s0)
    a3 = torch.where(s < s0, sig_s0 * (s / s0).clamp_min(1e-30) ** (s0 * (1.0 - sig_s0)), torch.sigmoid(s))
    b0 = (a3 ** gam) * 10.0 ** (a0 + a1 * v)
    b1 = c[15] * v * (0.5 + x - torch.tanh(4.0 * (x + c[16] + c[17] * v)))
    b1 = (c[14] * (1.0 + x) - b1) / (torch.exp(0.34 * (v - c[18])) + 1.0)
    v0 = c[21] + c[22] * x + c[23] * x2
    d1 = c[24] + c[25] * x + c[26] * x2
    d2 = c[27] + c[28] * x
    v2 = v / v0 + 1.0
    v2 = torch.where(v2 < y0, a + b * (v2 - 1.0).clamp_min(0) ** pn, v2)
    b2 = (-d1 + d2 * v2) * torch.exp(-v2)
    return b0 * (1.0 + b1 * cosphi + b2 * (2.0 * cosphi * cosphi - 1.0)) ** 1.6


def make_scene(lines, samples, total_lines, line0, seed, device, chunk=500):
    """SURVEY.md 8d generator: incidence ramp 30..46 deg, cyclone-like wind, ENL-100 speckle, ancillary =
    truth + smooth 1.5 m/s noise, 0.5 % NaN sigma0, first 8 samples NaN incidence.  float32/complex64."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    L, S = total_lines, samples
    inc = torch.empty((lines, samples), dtype=torch.float32, device=device)
    s_vv = torch.empty((lines, samples), dtype=torch.float32, device=device)
    anc = torch.empty((lines, samples), dtype=torch.complex64, device=device)
    # smooth ancillary noise: coarse 1/64 grid, bilinear upsampling
    ch, cw = lines // 64 + 2, samples // 64 + 2
    coarse = torch.randn((1, 2, ch, cw), generator=g, device=device, dtype=torch.float32) * 1.5
    noise = torch.nn.functional.interpolate(coarse, size=(lines, samples), mode="bilinear", align_corners=True)[0]
    ss = torch.arange(samples, device=device, dtype=torch.float64)[None, :]
    for l0 in range(0, lines, chunk):
        l1 = min(lines, l0 + chunk)
        ll = (torch.arange(l0, l1, device=device, dtype=torch.float64) + line0)[:, None]
        inc_c = 30.0 + 16.0 * ss / max(S - 1, 1) + 0.02 * torch.sin(2 * math.pi * ll / L)
        r2 = (ll - L / 2) ** 2 + (ss - S / 2) ** 2
        w_t = (9 + 6 * torch.sin(3 * math.pi * ll / L) * torch.cos(2 * math.pi * ss / S)
               + 12 * torch.exp(-r2 / (0.15 * min(L, S)) ** 2)).clamp(1, 40)
        dir_t = torch.atan2(ll - L / 2, ss - S / 2) + 0.6
        sig = cmod5n_torch(inc_c, w_t, torch.rad2deg(dir_t))
        speckle = torch._standard_gamma(torch.full(sig.shape, 100.0, device=device, dtype=torch.float32), generator=g) / 100.0
        sig32 = (sig.float() * speckle)
        holes = torch.rand(sig.shape, generator=g, device=device) < 0.005
        sig32[holes] = float("nan")
        inc32 = inc_c.float().expand(l1 - l0, samples).clone()
        inc32[:, :8] = float("nan")
        inc[l0:l1] = inc32
        s_vv[l0:l1] = sig32
        anc[l0:l1] = torch.complex((w_t * torch.cos(dir_t)).float() + noise[0, l0:l1],
                                   (w_t * torch.sin(dir_t)).float() + noise[1, l0:l1])
    return inc, s_vv, anc


# ------------------------------------------------------------------------------------------ helpers
def build_product_lut(resolution=None):
    from xsarsea_amd.windspeed import _engine, get_model
    kwargs = {} if resolution in (None, "high") else {"resolution": resolution}
    lut = get_model("gmf_cmod5n")._lut(units="dB", **kwargs)
    return lut, _engine._co_dict(lut)


def cpu_baseline_and_parity(ctx, inc, s_vv, anc, algo, budget_s=14.0):
    """Oracle (plain-C restatement of the reference kernel, reference LUT layout, all host cores) on a
    centre crop sized for ~budget_s; the same crop is re-run on the GPU and compared index by index."""
    from oracle import cport
    from oracle import invert as oinv
    from oracle import lut as olut
    from xsarsea_amd import _lib
    lco = olut.to_lut("gmf_cmod5n")
    prep = oinv.Prepared(lco, None)
    cores = cport.max_threads()
    lines, samples = inc.shape

    def crop(n_side):
        l0, s0 = max((lines - n_side) // 2, 0), max((samples - n_side) // 2, 0)
        sl = (slice(l0, l0 + n_side), slice(s0, s0 + n_side))
        return [t[sl].contiguous().cpu().numpy() for t in (inc, s_vv, anc)]

    def run_oracle(ci, cs, ca):
        nan = np.full(ci.shape, np.nan)
        t0 = time.perf_counter()
        res = cport.invert_numpy(prep, ci, oinv.to_db(cs), nan, nan, ca, nthreads=cores, return_idx=True,
                                 reference_layout=True)
        return res, time.perf_counter() - t0

    cal = int(min(lines, samples, max(64, 16 * math.isqrt(cores))))
    ci, cs, ca = crop(cal)
    run_oracle(ci, cs, ca)  # warms the LUT pages and the thread pool
    _, t_small = run_oracle(ci, cs, ca)
    per_px = t_small / ci.size
    side = int(max(32, min(min(lines, samples), math.sqrt(budget_s / per_px))))
    ci, cs, ca = crop(side)
    (o_co, _, o_idx), t_cpu = run_oracle(ci, cs, ca)
    cpu = {"value": round(ci.size / t_cpu / 1e6, 6), "unit": "Mpixels/s", "cores": cores, "kind": "port",
           "sample": f"{side}x{side} centre crop of the same raster ({ci.size} px, {t_cpu:.1f} s), "
                     f"oracle/invert_c.c with the reference's (wspd,phi,inc) LUT layout, OpenMP rows"}
    # GPU on the same crop: strict (host dB, as the reference computes it) and fused device dB
    ctx.synchronize()
    g_strict = ctx.invert_host(ci, sigma0_co=oinv.to_db(cs), anc=ca, sigma0_is_db=True, algo=algo, want_idx=True,
                               out_dtype=np.complex64)
    g_fused = ctx.invert_host(ci, sigma0_co=cs, anc=ca, algo=algo, want_idx=True, out_dtype=np.complex64)
    valid = o_idx[..., 0] >= 0

    def rel_err(g):
        ok = valid & np.all(g[2][..., :2] == o_idx[..., :2], axis=-1)
        if not ok.any():
            return None
        ref = o_co[ok]
        return float(np.max(np.abs(g[0][ok].astype(np.complex128) - ref) / np.maximum(np.abs(ref), 1e-3)))

    parity = {
        "crop_pixels": int(ci.size),
        "nan_mask_equal": bool(np.array_equal(np.isnan(g_fused[0].real), np.isnan(o_co.real))),
        "index_match_host_db": float(np.mean(np.all(g_strict[2][..., :2] == o_idx[..., :2], axis=-1))),
        "index_match_device_db": float(np.mean(np.all(g_fused[2][..., :2] == o_idx[..., :2], axis=-1))),
        "max_rel_err_uv_c64": rel_err(g_strict),
    }
    return cpu, parity


def bench_detrend(args, ctx, stream, s_vv, lines, samples):
    """`sigma0 / ratio_row` (detrend.py:63-64) on the resident float32 raster -> float64: 4 B read + 8 B written per pixel."""
    from xsarsea_amd import _lib
    det = torch.empty((lines, samples), dtype=torch.float64, device=s_vv.device)
    ratio = np.random.default_rng(0).uniform(0.5, 2.0, samples)

    def step():
        ctx.detrend_raw(lines, samples, _lib.XSW_F32, _lib.XSW_F64, _lib.MEM_DEVICE, s_vv.data_ptr(), ratio, det.data_ptr())

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    px = lines * samples
    achieved = 12.0 * px / (kernel_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(REPO, "profiles", "r01_f_hbm_traffic_summary.json")
    if os.path.exists(tfile) and (lines, samples) == (20000, 20000):
        try:  # k_detrend is the calibration kernel of the traffic passes: WRITE_SIZE is exact, FETCH_SIZE x the gfx950 factor 2
            raw = json.load(open(tfile))["raw_KiB"]["k_detrend"]
            traffic = int(raw["WRITE_SIZE"] * 1024 + 2.0 * raw["FETCH_SIZE"] * 1024)
        except Exception:
            traffic = None
    print(json.dumps({
        "metric": "Mpixels/s sigma0_detrend", "value": round(px * args.steps / dt / 1e6, 1), "unit": "Mpixels/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"sigma0_detrend kernel, {lines}x{samples} float32 sigma0 -> float64, one divisor per sample"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": "k_detrend",
                     "kernel_ms": round(kernel_ms, 4), "bytes_per_pixel": 12}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--lines", type=int, default=20000)
    ap.add_argument("--samples", type=int, default=20000)
    ap.add_argument("--algo", default="pruned", choices=["pruned", "exhaustive", "exhaustive_f64", "exact"])
    ap.add_argument("--mode", default="mono", choices=["mono", "dual", "detrend"],
                    help="mono: CMOD5.N VV (the metric's workload); dual: + Sentinel-1 VH GMF cross-pol refinement (config 3); "
                         "detrend: the sigma0_detrend kernel alone (the HBM-bound kernel of the path; N = 1 only)")
    ap.add_argument("--resolution", default="high", choices=["high", "low"],
                    help="LUT resolution passed to Model.to_lut: high = the reference default 501x499x181, low = 51x250x73 "
                         "(a parity-test configuration, SURVEY 8d; implies no CPU baseline / exhaustive figure)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stats", action="store_true", help="also report evaluated candidates per pixel (extra pass)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs (never set by the driver): XSW_BENCH_BACKEND=gloo and XSW_BENCH_ONE_DEVICE=1 let the whole N > 1
    # code path run as several ranks on a one-GPU box (everything but RCCL itself)
    backend = os.environ.get("XSW_BENCH_BACKEND", "nccl")
    if os.environ.get("XSW_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    n_gpus = world
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import xsarsea_amd
    from xsarsea_amd import _lib
    xsarsea_amd.options.device = local_rank  # one process per GPU: LUT preparation also runs on this rank's device
    lut, co_dict = build_product_lut(args.resolution)
    if args.resolution != "high":
        args.no_cpu_baseline = True
    ctx = _lib.Context(local_rank)
    stream = torch.cuda.Stream(device=device)  # the kernels, the events and RCCL all use this stream
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    ctx.upload_luts(co=co_dict)

    lines, samples = args.lines, args.samples
    inc, s_vv, anc = make_scene(lines, samples, lines * n_gpus, rank * lines, 20260320 + 2 + rank, device)
    if args.mode == "detrend":
        return bench_detrend(args, ctx, stream, s_vv, lines, samples)
    out = torch.empty((lines, samples), dtype=torch.complex64, device=device)
    s_vh = dsig = out_dual = None
    if args.mode == "dual":
        from xsarsea_amd.windspeed import _engine, get_model
        ctx.upload_luts(cr=_engine._cr_dict(get_model("gmf_s1_v2")._lut(units="dB")))
        g = torch.Generator(device=device)
        g.manual_seed(777 + rank)
        w_abs = anc.abs().clamp(3.0, 80.0).double()  # cross-pol truth ~ the a-priori speed (synthetic)
        from xsarsea_amd.windspeed import gmfs_impl  # scene synthesis only: coefficients of the S1 VH GMF
        _vh = gmfs_impl._VH_MODELS["gmf_s1_v2"]
        z1, z2, cc = _vh.z1, _vh.z2, _vh.logistic
        incd = inc.double().nan_to_num(35.0)
        sig1 = z1[0] * w_abs ** (z1[1] + z1[2] * incd)
        sig2 = (z2[0] + z2[1] * incd + z2[2] * incd ** 2) * w_abs ** (z2[3] + z2[4] * incd + z2[5] * incd ** 2)
        vh = sig1 * torch.sigmoid(cc[0] * (w_abs - cc[1])) + sig2 * torch.sigmoid(cc[2] * (w_abs - cc[3]))
        speck = torch._standard_gamma(torch.full(vh.shape, 100.0, device=device, dtype=torch.float32), generator=g) / 100.0
        s_vh = (vh.float() * speck + 10 ** -3.5).contiguous()
        dsig = ((1.25 / (s_vh / 10 ** -3.5)) ** 4.0).contiguous()
        out_dual = torch.empty((lines, samples), dtype=torch.complex64, device=device)
        del w_abs, incd, sig1, sig2, vh, speck
    full = None  # rank 0: the gathered (lines * N) x samples raster
    if world > 1 and rank == 0:
        full = torch.empty((lines * world, samples), dtype=torch.complex64, device=device)
    algo = _lib.ALGOS[args.algo]
    from xsarsea_amd import multi_gpu

    # N > 1: the tile is inverted in row chunks so that chunk k travels to rank 0 over xGMI while chunk k+1
    # is being inverted (RCCL runs on its own stream; requests are waited for at the end of the step)
    n_chunks = 8 if world > 1 else 1  # the last chunk's transfer is the only exposed one: 1/8 of a tile
    bounds = [(lines * c // n_chunks, lines * (c + 1) // n_chunks) for c in range(n_chunks)]
    es_in = 4   # float32 rasters
    pending = []

    def step_chunked():
        for (r0, r1) in bounds:
            off = r0 * samples
            ctx.invert_raw(r1 - r0, samples, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, inc.data_ptr() + off * es_in,
                           s_vv.data_ptr() + off * es_in, None, None, anc.data_ptr() + off * 8, out.data_ptr() + off * 8,
                           None, algo=algo)
            pending.extend(multi_gpu.gather_rows_async(out, lines * world, r0, r1, dst=0, out=full))

    def step():
        if world > 1 and args.mode == "mono":
            return step_chunked()
        if args.mode == "dual":
            ctx.invert_raw(lines, samples, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, inc.data_ptr(), s_vv.data_ptr(),
                           s_vh.data_ptr(), dsig.data_ptr(), anc.data_ptr(), out.data_ptr(), out_dual.data_ptr(),
                           algo=algo, dual_select=True)
        else:
            ctx.invert_raw(lines, samples, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, inc.data_ptr(), s_vv.data_ptr(),
                           None, None, anc.data_ptr(), out.data_ptr(), None, algo=algo)

    def gather():
        if world > 1 and args.mode == "mono":
            while pending:  # the single exchange of the path, started chunk by chunk inside step()
                pending.pop().wait()
        elif world > 1:
            multi_gpu.gather_rows(out, lines * world, dst=0, out=full)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
        gather()
    fence()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
        gather()
    fence()
    dt = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    stats = None
    if args.stats:
        ctx.stats_enable(True)
        step()
        gather()
        fence()
        stats = ctx.stats()
        ctx.stats_enable(False)

    if rank == 0:
        px_step = lines * samples * n_gpus
        value = px_step * args.steps / dt / 1e6
        bytes_px = (BYTES_READ_PX + BYTES_WRITE_PX) if args.mode == "mono" else (24 + 16)  # dual: +vh, +dsig; 2 outputs
        achieved = bytes_px * lines * samples / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(REPO, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                key = f"{args.algo}_{lines}x{samples}" if (args.mode == "mono" and args.resolution == "high") else "-"
                traffic = tj.get(key)
            except Exception:
                traffic = None
        res = {
            "metric": "Mpixels/s wind inversion (CMOD5.N, 20k x 20k sigma0)",
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"CMOD5.N {'mono-VV' if args.mode == 'mono' else 'dual-pol (VV + S1 VH GMF)'} inversion, {lines}x{samples} float32 sigma0/incidence + complex64 "
                                   f"ancillary per GPU, {'default' if args.resolution == 'high' else 'resolution=low'} LUT "
                                   f"{'x'.join(str(int(x)) for x in lut.shape)} ({int(lut.shape[1] * lut.shape[2])} candidates/pixel), "
                                   f"complex64 out, algo={args.algo}, mode={args.mode}",
                       "lines_per_gpu": lines, "samples": samples, "lut": [int(x) for x in lut.shape],
                       "parallelism": f"row tiles x{n_gpus}" + (", RCCL gather to rank 0 in the step" if n_gpus > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "kernel": {"exhaustive": "k_invert_exhaustive32", "exhaustive_f64": "k_invert_exhaustive"}.get(args.algo, "k_invert"),
                         "kernel_ms": round(kernel_ms, 3), "bytes_per_pixel": bytes_px,
                         "note": f"algorithmic raster bytes ({bytes_px} B read+written per pixel) / mean kernel time "
                                 "(HIP events on the launch stream); the search itself is VALU-issue bound, see valu"},
        }
        # the honest binding resource: float64 VALU issue
        cand_full = lut.shape[1] * lut.shape[2]
        res["valu"] = {"algorithmic_candidates_per_pixel": int(cand_full),
                       "algorithmic_Gcand_per_s": round(cand_full * lines * samples / (kernel_ms * 1e-3) / 1e9, 1)}
        if stats:
            res["valu"]["evaluated_candidates_per_pixel"] = round(stats["cand_co"] / max(stats["pixels_co"], 1), 1)
            res["valu"]["pixels_exact_fallback"] = stats["pixels_exact"]
        if n_gpus == 1 and args.mode == "mono" and args.algo == "pruned" and not args.no_cpu_baseline:
            # like-for-like figure: the literal exhaustive sweep (every one of the 90319 candidates scored per pixel,
            # LUT tiled through LDS, float32 screening + float64 settle) on the first lines of the same raster
            xl = max(4, min(lines, 2000))
            ex_alg = _lib.ALGOS["exhaustive"]

            def xstep():
                ctx.invert_raw(xl, samples, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, inc.data_ptr(), s_vv.data_ptr(),
                               None, None, anc.data_ptr(), out.data_ptr(), None, algo=ex_alg)
            xstep()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            xstep()
            e1.record(stream)
            torch.cuda.synchronize()
            xms = e0.elapsed_time(e1)
            res["exhaustive"] = {"value": round(xl * samples / (xms * 1e-3) / 1e6, 3), "unit": "Mpixels/s",
                                 "workload": f"first {xl} lines x {samples} samples of the same raster, every candidate scored",
                                 "kernel": "k_invert_exhaustive32", "kernel_ms": round(xms, 3),
                                 "Gcand_per_s": round(cand_full * xl * samples / (xms * 1e-3) / 1e9, 1)}
        if n_gpus == 1 and not args.no_cpu_baseline and args.mode == "mono":
            cpu, parity = cpu_baseline_and_parity(ctx, inc, s_vv, anc, args.algo)
            res["cpu_baseline"] = cpu
            res["parity"] = parity
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
