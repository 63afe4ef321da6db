#!/usr/bin/env python3
"""Benchmark of the wind-inversion hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config metric|2|3|4|5] [--scaling strong|weak] ...

One "step" = one pass of `xsw_invert` (default: CMOD5.N mono-VV, reference-default 0.1 m/s x 1 deg x 0.1 deg LUT)
over a synthetic float32 raster that is already resident in HBM (sigma0 + incidence + complex64 ancillary wind in,
complex64 wind out).  N = 1 workload: the 20000 x 20000 raster BASELINE.json's metric is quoted on.

N > 1: one process per GPU.  `python bench.py --gpus N` starts the N ranks itself (fresh child processes, spawned
before this process touches a GPU); under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`
the ranks are torchrun's.  Default `--scaling strong`: the SAME raster (20000 x 20000, or `--config 4`'s
25000 x 17000) is row-tiled over the ranks (`multi_gpu.tile_bounds`, the last rank takes the remainder) and the
output tiles are gathered on rank 0 over RCCL inside the timed step, chunk by chunk behind the kernel.
`--scaling weak`: every rank inverts a full-size tile of an N-times taller raster.

Prints ONE JSON line (rank 0): throughput, the HBM roofline of the dominant kernel (algorithmic bytes / HIP-event
kernel time) with the VALU view of the same kernel next to it, the sigma0_detrend kernel (the HBM-bound kernel of
the path), LUT build/upload time, the bit-parity configuration (sigma0 already in dB), and a CPU baseline (the
oracle's C restatement of the reference kernel on this box's cores, on a bounded crop of the same raster).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: required for multi-process RCCL on this host driver

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LANE_OPS_PEAK = 7.86e13  # 256 CU x 128 lanes x 2.4 GHz: the PACKED FP32 rate SURVEY.md 8d names (two lanes' worth per issue slot)
LANE_OPS_PEAK_ISSUE = 3.93e13  # 256 CU x 4 SIMDs x 16 lanes per clock x 2.4 GHz: one wave64 instruction per 4 cycles per SIMD -- what a float64 (non-packed) search can issue
OPS_PER_CANDIDATE = 6    # sub, scale, square-accumulate, separable wind term, compare, select
BYTES_READ_PX = 16       # inc f32 + sigma0 f32 + ancillary complex64   
BYTES_WRITE_PX = 8       # complex64 wind
PROFILE_ROUND = "r05"
PCIE_PEAK_GBS = 64.0      # PCIe Gen5 x16, one direction (MI355X_MICROARCH.md host link)

CONFIGS = {  # BASELINE.json configs (1 is the CPU plumbing case: tests/test_gpu_api.py::test_sigma0_detrend)
    "metric": dict(lines=20000, samples=20000, mode="mono", lut="cmod5n", note="the metric's raster"),
    "2": dict(lines=10000, samples=10000, mode="mono", lut="cmod5n", note="config 2"),
    "3": dict(lines=20000, samples=20000, mode="dual", lut="cmod5n", note="config 3"),
    "4": dict(lines=25000, samples=17000, mode="mono", lut="cmod5n", note="config 4 (S1 IW full swath)"),
    "5": dict(lines=20000, samples=20000, mode="mono", lut="cmod7", note="config 5 (CMOD7-shaped table)"),
}


# ------------------------------------------------------------------------------------------ synthetic scene
_C5N = [0.0, -0.6878, -0.7957, 0.338, -0.1728, 0.0, 0.004, 0.1103, 0.0159, 6.7329, 2.7713, -2.2885, 0.4971, -0.725,
        0.045, 0.0066, 0.3222, 0.012, 22.7, 2.0813, 3.0, 8.3659, -3.3428, 1.3236, 6.2437, 2.3893, 0.3249, 4.159, 1.693]


def cmod5n_torch(inc, v, phi_deg):
    """CMOD5.N forward model with torch ops (float64): the check of `cmod5n_device` in the gmf_eval figure, and scene synthesis
    on a CPU device."""
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


def cmod5n_device(inc, v, phi_deg):
    """CMOD5.N through the PRODUCT's forward-GMF kernel (xsw_gmf_eval, the device form of GmfModel.__call__(broadcast=True)):
    float64 tensors of one shape on a GPU -> sigma0 (linear).  The bench scene is generated with it (dogfood; untimed)."""
    from xsarsea_amd import _lib
    shape = torch.broadcast_shapes(inc.shape, v.shape, phi_deg.shape)
    a, b, c = (t.to(torch.float64).expand(shape).contiguous() for t in (inc, v, phi_deg))
    out = torch.empty(shape, dtype=torch.float64, device=a.device)
    ctx = _lib.default_context(a.device.index if a.device.index is not None else torch.cuda.current_device())
    with ctx.lock:
        ctx.set_stream(torch.cuda.current_stream(a.device).cuda_stream)
        try:
            ctx.gmf_eval_raw(_lib.GMF_IDS["gmf_cmod5n"], out.numel(), _lib.MEM_DEVICE, a.data_ptr(), b.data_ptr(), c.data_ptr(), out.data_ptr())
        finally:
            ctx.use_own_stream()
    for t in (a, b, c):
        t.record_stream(torch.cuda.current_stream(a.device))
    return out


def make_scene(lines, samples, total_lines, line0, seed, device, chunk=500, inc_range=(30.0, 46.0), anc_scale=1.0, outlier_frac=0.0):
    """SURVEY.md 8d generator: incidence ramp 30..46 deg, cyclone-like wind, ENL-100 speckle, ancillary =
    truth + smooth 1.5 m/s noise, 0.5 % NaN sigma0, first 8 samples NaN incidence.  float32/complex64.
    inc_range / anc_scale: the hard scenes of the bench line (near-range incidences where CMOD5.N saturates and turns over
    inside the search windows; an a-priori wind that is `anc_scale` times the truth).  outlier_frac: that share of the pixels,
    in blobs of 16 x 16 pixels, has its sigma0 multiplied by 10 (+10 dB; half of the blobs) or by 31.6 (+15 dB): ships, land, ice,
    rain cells -- sigma0 that no wind near the a-priori one explains."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    L, S = total_lines, samples
    inc = torch.empty((lines, samples), dtype=torch.float32, device=device)
    s_vv = torch.empty((lines, samples), dtype=torch.float32, device=device)
    anc = torch.empty((lines, samples), dtype=torch.complex64, device=device)
    # smooth ancillary noise: coarse 1/64 grid, bilinear upsampling
    ch, cw = lines // 64 + 2, samples // 64 + 2
    coarse = torch.randn((1, 2, ch, cw), generator=g, device=device, dtype=torch.float32) * 1.5
    noise = torch.nn.functional.interpolate(coarse, size=(max(lines, 1), samples), mode="bilinear", align_corners=True)[0]
    ss = torch.arange(samples, device=device, dtype=torch.float64)[None, :]
    gain = None
    if outlier_frac > 0.0:
        bh, bw = lines // 16 + 1, samples // 16 + 1
        u = torch.rand((bh, bw), generator=g, device=device)
        gb = torch.where(u < 0.5 * outlier_frac, 10.0, torch.where(u < outlier_frac, 31.6, 1.0)).float()
        gain = gb.repeat_interleave(16, 0).repeat_interleave(16, 1)[:lines, :samples]
    for l0 in range(0, lines, chunk):
        l1 = min(lines, l0 + chunk)
        ll = (torch.arange(l0, l1, device=device, dtype=torch.float64) + line0)[:, None]
        inc_c = inc_range[0] + (inc_range[1] - inc_range[0]) * ss / max(S - 1, 1) + 0.02 * torch.sin(2 * math.pi * ll / L)
        r2 = (ll - L / 2) ** 2 + (ss - S / 2) ** 2
        w_t = (9 + 6 * torch.sin(3 * math.pi * ll / L) * torch.cos(2 * math.pi * ss / S)
               + 12 * torch.exp(-r2 / (0.15 * min(L, S)) ** 2)).clamp(1, 40)
        dir_t = torch.atan2(ll - L / 2, ss - S / 2) + 0.6
        sig = cmod5n_device(inc_c, w_t, torch.rad2deg(dir_t)) if device.type == "cuda" else cmod5n_torch(inc_c, w_t, torch.rad2deg(dir_t))
        speckle = torch._standard_gamma(torch.full(sig.shape, 100.0, device=device, dtype=torch.float32), generator=g) / 100.0
        sig32 = (sig.float() * speckle)
        if gain is not None:
            sig32 = sig32 * gain[l0:l1]
        holes = torch.rand(sig.shape, generator=g, device=device) < 0.005
        sig32[holes] = float("nan")
        inc32 = inc_c.float().expand(l1 - l0, samples).clone()
        inc32[:, :8] = float("nan")
        inc[l0:l1] = inc32
        s_vv[l0:l1] = sig32
        anc[l0:l1] = torch.complex((w_t * torch.cos(dir_t)).float() * anc_scale + noise[0, l0:l1],
                                   (w_t * torch.sin(dir_t)).float() * anc_scale + noise[1, l0:l1])
    return inc, s_vv, anc


def make_crosspol(inc, anc, seed, device):
    """Cross-pol rasters of the dual-pol workload: sigma0_vh = S1 VH GMF(inc, |ancillary|) x speckle + NESZ (10^-3.5),
    dsig_cr = (1.25 / (sigma0_vh / nesz))^4 (windspeed/utils.py:82-86 of the reference).  float32."""
    from xsarsea_amd.windspeed import gmfs_impl  # scene synthesis only: coefficients of the S1 VH GMF
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    w_abs = anc.abs().clamp(3.0, 80.0).double()  # cross-pol truth ~ the a-priori speed (synthetic)
    _vh = gmfs_impl._VH_MODELS["gmf_s1_v2"]
    z1, z2, cc = _vh.z1, _vh.z2, _vh.logistic
    incd = inc.double().nan_to_num(35.0)
    sig1 = z1[0] * w_abs ** (z1[1] + z1[2] * incd)
    sig2 = (z2[0] + z2[1] * incd + z2[2] * incd ** 2) * w_abs ** (z2[3] + z2[4] * incd + z2[5] * incd ** 2)
    vh = sig1 * torch.sigmoid(cc[0] * (w_abs - cc[1])) + sig2 * torch.sigmoid(cc[2] * (w_abs - cc[3]))
    speck = torch._standard_gamma(torch.full(vh.shape, 100.0, device=device, dtype=torch.float32), generator=g) / 100.0
    s_vh = (vh.float() * speck + 10 ** -3.5).contiguous()
    dsig = ((1.25 / (s_vh / 10 ** -3.5)) ** 4.0).contiguous()
    return s_vh, dsig


# ------------------------------------------------------------------------------------------ helpers
def cmod7_shaped_model(tmpdir):
    """BASELINE config 5: the real CMOD7 table is not distributable, so a CMOD7-FORMAT file (250 x 73 x 51 float32,
    Fortran order, record markers; cmod7.py:27-40) is synthesised from CMOD5.N x (1 + 5 % smooth perturbation) and
    read back through the product's own reader + low->high interpolation."""
    from xsarsea_amd.windspeed import cmod7, gmfs_impl
    w, p, i = np.arange(1, 251) * 0.2, np.arange(73) * 2.5, np.arange(16, 67) * 1.0
    table = gmfs_impl._cmod5_sigma0(gmfs_impl._CMOD5N, i[None, None, :], w[:, None, None], p[None, :, None])
    table = (table * (1 + 0.05 * np.sin(w[:, None, None] / 7.0) * np.cos(np.radians(p[None, :, None])))).astype(np.float32)
    os.makedirs(tmpdir, exist_ok=True)
    cmod7.write_cmod7_table(os.path.join(tmpdir, "gmf_cmod7_vv.dat_little_endian"), table)
    return cmod7.register_cmod7(tmpdir)


def build_product_lut(resolution=None, which="cmod5n", timings=None):
    from xsarsea_amd.windspeed import _engine, get_model
    kwargs = {} if resolution in (None, "high") else {"resolution": resolution}
    if which == "cmod7":
        import tempfile
        model = cmod7_shaped_model(os.path.join(tempfile.gettempdir(), f"xsw_cmod7_{os.getpid()}"))
    else:
        model = get_model("gmf_cmod5n")
    t0 = time.perf_counter()
    lut = model._lut(units="dB", **kwargs)
    if timings is not None:
        timings["lut_build_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    return lut, _engine._co_dict(lut)


def cpu_baseline_and_parity(ctx, inc, s_vv, anc, algo, budget_s=14.0):
    """Oracle (plain-C restatement of the reference kernel, reference LUT layout, all host cores) on a
    centre crop sized for ~budget_s; the same crop is re-run on the GPU and compared index by index."""
    from oracle import cport
    from oracle import invert as oinv
    from oracle import lut as olut
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
    # the numpy restatement beside it (same temporaries, hence the same memory behaviour, as the reference's
    # per-pixel numpy expressions; one core -- numba's thread pool is what the C port's OpenMP rows stand for)
    nside = int(max(8, min(side, math.isqrt(max(int(2.0 / 2e-3), 64)))))  # ~2 s at ~2 ms per pixel
    l0n, s0n = (ci.shape[0] - nside) // 2, (ci.shape[1] - nside) // 2
    sub = (slice(l0n, l0n + nside), slice(s0n, s0n + nside))
    nan_s = np.full((nside, nside), np.nan)
    t0 = time.perf_counter()
    n_co, _, n_idx = oinv.invert_numpy(prep, ci[sub], oinv.to_db(cs[sub]), nan_s, nan_s, ca[sub], return_idx=True)
    t_np = time.perf_counter() - t0
    cpu["numpy_restatement"] = {"value": round(nside * nside / t_np / 1e6, 6), "unit": "Mpixels/s", "cores": 1,
                                "sample": f"{nside}x{nside} centre of that crop ({t_np:.1f} s), oracle/invert.py (numpy, float64 temporaries of 499x181 per pixel)",
                                "equals_c_port": bool(np.array_equal(n_idx[..., :2], o_idx[sub][..., :2]))}
    # GPU on the same crop: strict (host dB, as the reference computes it) and fused device dB
    ctx.synchronize()
    g_strict = ctx.invert_host(ci, sigma0_co=oinv.to_db(cs), anc=ca, sigma0_is_db=True, algo=algo, want_idx=True,
                               out_dtype=np.complex64)
    g_fused = ctx.invert_host(ci, sigma0_co=cs, anc=ca, algo=algo, want_idx=True, out_dtype=np.complex64)
    valid = o_idx[..., 0] >= 0

    def rel_err(g, matched_only):
        """max over the valid pixels of |(u,v)_gpu - (u,v)_ref| / max(|ref|, 1e-3); matched_only: pixels on the same grid point"""
        ok = valid & (np.all(g[2][..., :2] == o_idx[..., :2], axis=-1) if matched_only else True)
        if not ok.any():
            return None, 0
        ref = o_co[ok]
        err = np.abs(g[0][ok].astype(np.complex128) - ref) / np.maximum(np.abs(ref), 1e-3)
        return float(np.max(err)), int(np.sum(err > 1e-4))

    e_strict, out_strict = rel_err(g_strict, False)
    e_fused, out_fused = rel_err(g_fused, False)
    e_fused_same, _ = rel_err(g_fused, True)
    parity = {
        "crop_pixels": int(ci.size), "valid_pixels": int(valid.sum()),
        "nan_mask_equal": bool(np.array_equal(np.isnan(g_fused[0].real), np.isnan(o_co.real))),
        # the bit-parity route (sigma0 converted to dB by numpy on the host, as the reference does)
        "index_match_host_db": float(np.mean(np.all(g_strict[2][..., :2] == o_idx[..., :2], axis=-1))),
        "max_rel_err_uv_c64": e_strict, "pixels_outside_1e-4_host_db": out_strict,
        # the TIMED route (`value`: sigma0 -> dB fused on the device)
        "index_match_device_db": float(np.mean(np.all(g_fused[2][..., :2] == o_idx[..., :2], axis=-1))),
        "max_rel_err_uv_c64_device_db": e_fused, "pixels_outside_1e-4_device_db": out_fused,
        "frac_outside_1e-4_device_db": round(out_fused / max(int(valid.sum()), 1), 8),
        "max_rel_err_uv_c64_device_db_same_grid_point": e_fused_same,
        "note": "device_db: the float32 log10 of the fused conversion is correctly rounded, numpy's (the reference's) is a few-ulp SIMD routine; "
                "the pixels_outside_1e-4_device_db pixels are near-ties that land one grid step (0.1 m/s or 1 deg) away",
        "lut_interp": "unpinned vs xarray.interp (restated as sequential scipy interp1d; DESIGN.md 6)",
    }
    return cpu, parity


def _profile_json(name):
    try:
        with open(os.path.join(REPO, "profiles", name)) as f:
            return json.load(f)
    except Exception:
        return None


_LOADED_CODE_SHA = []


def loaded_code_sha():
    """SHA-256 of the device code (.hip_fatbin) of the libxsw.so this process runs: counter figures measured on other code
    are not reported (profiles/collect_r03.sh stamps every summary with the hash it was measured on)."""
    if not _LOADED_CODE_SHA:
        try:
            from xsarsea_amd import _build
            _LOADED_CODE_SHA.append(_build.code_object_sha256())
        except Exception:
            _LOADED_CODE_SHA.append(None)
    return _LOADED_CODE_SHA[0]


def fresh_profile(name):
    """(summary, provenance) of a committed counter summary if it was measured on the loaded library's device code, else
    (None, {"stale_profile": why})."""
    tj = _profile_json(name)
    if not tj:
        return None, {"stale_profile": f"profiles/{name} absent"}
    on = tj.get("measured_on") or {}
    mine = loaded_code_sha()
    if not on.get("code_sha256") or mine is None or on["code_sha256"] != mine:
        return None, {"stale_profile": f"profiles/{name} was measured on device code {str(on.get('code_sha256'))[:16]} (commit {on.get('commit')}), "
                                       f"this run loads {str(mine)[:16]}: counters not reported"}
    return tj, {"measured_on": on}


def time_steps(step, steps, warmup, stream, after=None):
    """W untimed + K timed calls of step(); returns (wall seconds of the K steps, mean HIP-event ms of step())."""
    for _ in range(warmup):
        step()
        if after:
            after()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record(stream)
        step()
        b.record(stream)
        if after:
            after()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt, float(np.mean([a.elapsed_time(b) for a, b in ev]))


def end_to_end_parity_figure(ctx, _lib, inc, s_vv, anc, out, lines, samples, algo):
    """The bit-parity route with NOTHING left outside the timer: the linear float32 sigma0 raster starts on the HOST (pageable
    numpy), incidence / ancillary wind / output stay resident, as in `value`.  Timed = one `xsw_invert` call with
    XSW_MEM_DEVICE_SIGMA0_HOST: the library's workers take sigma0 in row chunks, the staging callback converts each chunk to dB
    with numpy's own float32 log10 (the reference's arithmetic, windspeed.py:126-130) straight into page-locked memory, the
    chunk is uploaded and searched on the worker's stream while other chunks are in other phases."""
    import ctypes
    s_host = s_vv.cpu().numpy()
    flat = s_host.reshape(-1)

    def stage(which, px0, npx, dst):
        if which != _lib.STAGE_SIGMA0_CO:
            return 0
        o = np.frombuffer((ctypes.c_char * (npx * 4)).from_address(dst), dtype=np.float32)
        with np.errstate(all="ignore"):
            np.add(flat[px0:px0 + npx], np.float32(1e-15), out=o)
            np.log10(o, out=o)
            np.multiply(o, np.float32(10), out=o)
        return 1

    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.invert_raw(lines, samples, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE_SIGMA0_HOST, inc.data_ptr(), s_host.ctypes.data, None, None,
                       anc.data_ptr(), out.data_ptr(), None, algo=algo, sigma0_is_db=True, stage=stage)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    px = lines * samples
    return {"value": round(px / best / 1e6, 1), "unit": "Mpixels/s", "ms": round(best * 1e3, 2),
            "bytes_over_pcie": 4 * px, "pcie_GBps": round(4 * px / best / 1e9, 1),
            "note": "host numpy float32 log10 in the staging step of the library's worker ring + upload of the dB chunks + kernels, one synchronous "
                    "call, all timed; best of 3 (the first call pins the staging buffers)"}


def host_path_figure(inc, s_vv, anc, samples, lines_host=5000):
    """The public drop-in call on HOST rasters: `invert_from_model(numpy float32 -> numpy complex128)` on the first 5000 lines of
    the same scene (1e8 pixels at 20000 samples), default options (float32 dB by numpy on the host: bit parity).  Everything is
    timed: dB pass, page-locked staging, PCIe both ways (16 B/px up, 4 B/px of grid codes down), kernels, host expansion."""
    import warnings
    import xsarsea_amd
    from xsarsea_amd import windspeed
    lh = min(lines_host, inc.shape[0])
    h_inc, h_s, h_anc = (t[:lh].contiguous().cpu().numpy() for t in (inc, s_vv, anc))
    px = lh * samples
    times = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = None
        for _ in range(5):  # the first call builds / uploads the LUT of the API's own context and pins the staging ring
            del res  # releasing the previous 1.6 GB result is the caller's business, not part of the call
            t0 = time.perf_counter()
            res = windspeed.invert_from_model(h_inc, h_s, ancillary_wind=h_anc, model="gmf_cmod5n")
            times.append(time.perf_counter() - t0)
    best = min(times[1:])
    up, down = 16 * px, 4 * px
    fig = {"workload": f"windspeed.invert_from_model(numpy float32 {lh}x{samples} -> numpy complex128), default options, pageable host arrays",
           "value": round(px / best / 1e6, 1), "unit": "Mpixels/s", "ms": round(best * 1e3, 2), "first_call_ms": round(times[0] * 1e3, 1),
           "bytes_over_pcie": up + down,
           "roofline": {"bound": "pcie", "achieved": round((up + down) / best / 1e9, 2), "peak": PCIE_PEAK_GBS, "unit": "GB/s",
                        "frac": round((up + down) / best / 1e9 / PCIE_PEAK_GBS, 4),
                        "note": "16 B/px up (incidence, dB sigma0, ancillary wind) + 4 B/px down (grid codes; the complex128 raster is expanded on the host)"},
           "host_threads": xsarsea_amd.options.host_threads or int(os.environ.get("XSW_HOST_THREADS", "12"))}
    del res
    return fig


def host_path_all_devices_figure(inc, s_vv, anc, samples, n_dev, lines_per_dev=2500):
    """N > 1 line: the drop-in call on HOST rasters with `xsarsea_amd.options.devices` = every GPU of the node, from ONE process
    (rank 0; the other ranks wait at the barrier): row tiles, one context and one host thread per GPU, no exchange.  Weak
    scaling by construction -- every GPU has its own PCIe link -- until the host's memory bandwidth runs out."""
    import warnings
    import xsarsea_amd
    from xsarsea_amd import windspeed
    one_dev = os.environ.get("XSW_BENCH_ONE_DEVICE") == "1"
    reps = -(-lines_per_dev * n_dev // inc.shape[0])
    h_inc, h_s, h_anc = (np.concatenate([t.contiguous().cpu().numpy()] * reps)[:lines_per_dev * n_dev] for t in (inc, s_vv, anc))
    px = h_inc.size
    prev = xsarsea_amd.options.devices
    xsarsea_amd.options.devices = [0] * n_dev if one_dev else list(range(n_dev))
    try:
        times = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = None
            for _ in range(4):
                del res
                t0 = time.perf_counter()
                res = windspeed.invert_from_model(h_inc, h_s, ancillary_wind=h_anc, model="gmf_cmod5n")
                times.append(time.perf_counter() - t0)
        best = min(times[1:])
        return {"workload": f"windspeed.invert_from_model(numpy float32 {h_inc.shape[0]}x{samples} -> numpy complex128), options.devices = {n_dev} GPUs, one process",
                "value": round(px / best / 1e6, 1), "unit": "Mpixels/s", "ms": round(best * 1e3, 2), "first_call_ms": round(times[0] * 1e3, 1),
                "pcie_GBps_per_device": round(20 * px / n_dev / best / 1e9, 2),
                "note": "16 B/px up + 4 B/px of grid codes down per GPU link; unmeasured on a multi-GPU node before this run"}
    except Exception as exc:  # a figure of the line, never a reason to lose the line
        return {"error": f"{type(exc).__name__}: {exc}"}
    finally:
        xsarsea_amd.options.devices = prev


def hard_scene_figures(ctx, _lib, stream, device, samples, algo, lines_hs=4000):
    """The benchmark scene is the friendly case for an exact branch-and-bound: its a-priori wind is the truth + 1.5 m/s, so the
    bound is tight and ~50 of 90 319 candidates are scored.  Three scenes where it is not, on a 4000-line band through the
    cyclone of the same generator (always reported, never part of `value`): the a-priori wind scaled by 0.6 (the feasible
    set is an arc of the sigma0 contour: hundreds of candidates per pixel genuinely score below the bound) and by 1.6 (the
    windows reach into the saturated rows of CMOD5.N), and incidences 17..33 deg (CMOD5.N saturates and turns over inside
    many search windows).  Results stay exact (tests, campaigns; profiles/hard_scenes.py --verify compares ten such scenes
    with the exhaustive sweep)."""
    out = {}
    lines_hs = int(lines_hs)
    o = torch.empty((lines_hs, samples), dtype=torch.complex64, device=device)
    for key, inc_range, scale, outl, what in (
            ("friendly_band", (30.0, 46.0), 1.0, 0.0, "the benchmark scene's own lines 8000..12000 (through the cyclone)"),
            ("sigma0_outliers_1pct", (30.0, 46.0), 1.0, 0.01, "the same lines with 1 % of the pixels (16 x 16 blobs) at sigma0 x 10 / x 31.6 (+10 / +15 dB: ships, land, rain cells)"),
            ("sigma0_outliers_5pct", (30.0, 46.0), 1.0, 0.05, "the same lines with 5 % of the pixels at sigma0 x 10 / x 31.6"),
            ("ancillary_x0.6", (30.0, 46.0), 0.6, 0.0, "a-priori wind = 0.6 x truth + noise, incidence 30..46 deg"),
            ("ancillary_x0.3", (30.0, 46.0), 0.3, 0.0, "a-priori wind = 0.3 x truth + noise, incidence 30..46 deg"),
            ("ancillary_x1.6", (30.0, 46.0), 1.6, 0.0, "a-priori wind = 1.6 x truth + noise, incidence 30..46 deg (windows into the GMF's saturated rows)"),
            ("ancillary_x2.5", (30.0, 46.0), 2.5, 0.0, "a-priori wind = 2.5 x truth + noise, incidence 30..46 deg"),
            ("incidence_17_33", (17.0, 33.0), 1.0, 0.0, "incidence 17..33 deg, a-priori wind = truth + noise"),
            ("incidence_17_33_x1.6", (17.0, 33.0), 1.6, 0.0, "incidence 17..33 deg, a-priori wind = 1.6 x truth + noise")):
        inc, s_vv, anc = make_scene(lines_hs, samples, 20000, 8000, 20260320 + 7, device, inc_range=inc_range, anc_scale=scale, outlier_frac=outl)
        torch.cuda.synchronize()

        def run():
            ctx.invert_raw(lines_hs, samples, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, inc.data_ptr(), s_vv.data_ptr(), None, None,
                           anc.data_ptr(), o.data_ptr(), None, algo=algo)
        run()
        ctx.synchronize()
        ctx.timing_enable(True)
        run()
        run()
        tm = ctx.timing()
        ctx.timing_enable(False)
        ctx.stats_enable(True)
        run()
        st = ctx.stats()
        ctx.stats_enable(2)  # the production chain, its kernels counting what they score
        run()
        ch = ctx.stats_chain()
        ctx.stats_enable(False)
        nl = max(tm["launches"], 1)
        per_s = lambda cnt, ms_k: None if not ms_k else float(f"{cnt / (ms_k * 1e-3):.3g}")
        ms = (tm["first_kernel_ms"] + tm["second_kernel_ms"] + tm["band2_kernel_ms"] + tm["blocks_kernel_ms"]) / max(tm["launches"], 1)
        out[key] = {"scene": what, "pixels": lines_hs * samples, "value": round(lines_hs * samples / ms / 1e3, 1), "unit": "Mpixels/s",
                    "k_invert_band_ms": round(tm["first_kernel_ms"] / max(tm["launches"], 1), 2),
                    "k_invert_band2_ms": round(tm["band2_kernel_ms"] / max(tm["launches"], 1), 2),
                    "k_invert_blocks_ms": round(tm["blocks_kernel_ms"] / max(tm["launches"], 1), 2),
                    "k_invert_list_ms": round(tm["second_kernel_ms"] / max(tm["launches"], 1), 2),
                    "pixels_to_band2_frac": round(tm["last_band2_pixels"] / (lines_hs * samples), 5),
                    "pixels_to_blocks_frac": round(tm["last_blocks_pixels"] / (lines_hs * samples), 5),
                    "pixels_left_to_the_list_frac": round(tm["last_list_pixels"] / (lines_hs * samples), 5),
                    "evaluated_candidates_per_pixel": round(st["cand_co"] / max(st["pixels_co"], 1), 1),
                    "pixels_exact_fallback": int(st["pixels_exact"]),
                    # measured on the production chain (xsw_stats_enable(ctx, 2)): what the kernels behind k_invert_band score, and how fast
                    "scored_candidates": {"k_invert_band2": ch["cand_band2"], "k_invert_blocks": ch["cand_blocks"], "k_invert_list": ch["cand_list"],
                                          "band2_records_refined_frac": round(ch["pixels_refined"] / max(tm["last_band2_pixels"], 1), 4)},
                    "scored_candidates_per_s": {"k_invert_band2": per_s(ch["cand_band2"], tm["band2_kernel_ms"] / nl),
                                                "k_invert_blocks": per_s(ch["cand_blocks"], tm["blocks_kernel_ms"] / nl),
                                                "k_invert_list": per_s(ch["cand_list"], tm["second_kernel_ms"] / nl),
                                                "chain_statistics_instantiation": per_s(st["cand_co"], ms)}}
        del inc, s_vv, anc
    return out


def detrend_figures(args, ctx, stream, s_vv, lines, samples):
    """`sigma0 / ratio_row` (detrend.py:63-64) on the resident float32 raster -> float64: 4 B read + 8 B written per
    pixel -- the one HBM-bound kernel of the path."""
    from xsarsea_amd import _lib
    det = torch.empty((lines, samples), dtype=torch.float64, device=s_vv.device)
    ratio = np.random.default_rng(0).uniform(0.5, 2.0, samples)

    def step():
        ctx.detrend_raw(lines, samples, _lib.XSW_F32, _lib.XSW_F64, _lib.MEM_DEVICE, s_vv.data_ptr(), ratio, det.data_ptr())

    dt, kernel_ms = time_steps(step, args.steps, max(args.warmup, 1), stream)
    del det
    px = lines * samples
    achieved = 12.0 * px / (kernel_ms * 1e-3) / 1e9
    traffic = None
    tj, prov = fresh_profile(f"{PROFILE_ROUND}_hbm_traffic_summary.json")
    if tj and (lines, samples) == (20000, 20000):
        try:  # k_detrend is the calibration kernel of the traffic passes: WRITE_SIZE is exact, FETCH_SIZE x the gfx950 factor 2
            raw = tj["raw_KiB"]["k_detrend"]
            traffic = int(raw["WRITE_SIZE"] * 1024 + 2.0 * raw["FETCH_SIZE"] * 1024)
        except Exception:
            traffic = None
    return {"workload": f"sigma0_detrend kernel, {lines}x{samples} float32 sigma0 -> float64, one divisor per sample",
            "value": round(px * args.steps / dt / 1e6, 1), "unit": "Mpixels/s", "kernel": "k_detrend",
            "kernel_ms": round(kernel_ms, 4), "bytes_per_pixel": 12,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_of_measured_copy_rate_6290": round(achieved / 6290.0, 4),
                         "traffic": traffic, "traffic_provenance": prov}}


def gmf_eval_figures(args, ctx, stream, device, n=400_000_000):
    """`xsw_gmf_eval(CMOD5.N)` -- the forward GMF of GmfModel.__call__(broadcast=True) (gmfs.py:293-316) -- on n already-broadcast
    float64 elements resident in HBM: 24 B read + 8 B written per element; ~1.5e3 float64 lane-instructions per element (seven
    pow, four exp, tanh, cos), so the kernel is VALU-bound: both rooflines are given."""
    from xsarsea_amd import _lib
    g = torch.Generator(device=device)
    g.manual_seed(3)
    inc = torch.rand(n, generator=g, device=device, dtype=torch.float64) * 30 + 17
    v = torch.rand(n, generator=g, device=device, dtype=torch.float64) * 40 + 0.5
    phi = torch.rand(n, generator=g, device=device, dtype=torch.float64) * 360 - 180
    out = torch.empty(n, dtype=torch.float64, device=device)
    gid = _lib.GMF_IDS["gmf_cmod5n"]

    def step():
        ctx.gmf_eval_raw(gid, n, _lib.MEM_DEVICE, inc.data_ptr(), v.data_ptr(), phi.data_ptr(), out.data_ptr())

    _, ms = time_steps(step, max(2, args.steps // 4), 1, stream)
    k = 1 << 20
    ref = cmod5n_torch(inc[:k], v[:k], phi[:k])
    rel = float(((out[:k] - ref).abs() / ref.abs()).max().item())
    del inc, v, phi, out
    achieved = 32.0 * n / (ms * 1e-3) / 1e9
    return {"workload": f"xsw_gmf_eval(gmf_cmod5n) on {n} broadcast float64 elements (incidence 17..47, wind 0.5..40.5, all directions)",
            "value": round(n / (ms * 1e-3) / 1e6, 1), "unit": "Melements/s", "kernel": "k_gmf_eval<CMOD5N>", "kernel_ms": round(ms, 3),
            "bytes_per_element": 32, "max_rel_diff_vs_torch_float64": float(f"{rel:.2e}"),
            "roofline": {"bound": "valu", "hbm": {"achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4)},
                         "note": "float64 transcendental arithmetic (ocml pow / exp / tanh / cos, ~1.5e3 lane-instructions per element), 119 VGPRs, no scratch "
                                 "(one kernel per model since round 5; round 4's one-for-all kernel: 181 VGPRs + 24 B of scratch)"}}


def latency_figures(local_rank):
    """What xsarsea is run at operationally: ~1 km grids, 1e5..1e6 pixels per call.  `invert_from_model` end to end on float32 numpy
    rasters (complex128 out), CMOD5.N default LUT, for both LUT routes (`options.lut_build`): the FIRST call with nothing
    prepared (fresh libxsw context, no memoised LUT: LUT preparation + install + staging start-up + search) and the WARM call
    (median of 7), with the warm call's phases.  Reference: the LUT is rebuilt on every call (windspeed.py:144)."""
    import xsarsea_amd
    from xsarsea_amd import _lib, options, windspeed
    from xsarsea_amd.windspeed import _engine, get_model, gmfs_impl
    out = {}
    saved = (options.lut_build, options.device)
    options.device = local_rank
    try:
        for lines, samples in ((250, 400), (1000, 1000)):
            rng = np.random.default_rng(lines)
            inc = np.broadcast_to(np.linspace(30, 46, samples, dtype=np.float32), (lines, samples)).copy()
            wt, pt = rng.uniform(2, 25, (lines, samples)), rng.uniform(-180, 180, (lines, samples))
            s_vv = (gmfs_impl._cmod5_sigma0(gmfs_impl._CMOD5N, inc.astype(np.float64), wt, pt) * rng.gamma(100, 0.01, (lines, samples))).astype(np.float32)
            anc = (wt * np.exp(1j * np.deg2rad(pt)) + rng.normal(0, 1.5, (lines, samples)) + 1j * rng.normal(0, 1.5, (lines, samples))).astype(np.complex64)
            key = f"{lines}x{samples}"
            out[key] = {"pixels": lines * samples}
            for route in ("host", "device"):
                options.lut_build = route
                model = get_model("gmf_cmod5n")
                model._lut_cache.clear()
                model.__dict__.pop("_device_luts", None)
                with _lib._default_ctx_lock:  # a fresh context: what the first call of a process pays (HIP itself is up already)
                    old = _lib._default_ctx.pop((int(local_rank), 0), None)
                if old is not None:
                    old.close()
                call = lambda: windspeed.invert_from_model(inc, s_vv, ancillary_wind=anc, model="gmf_cmod5n")
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    t0 = time.perf_counter()
                    call()
                    first_ms = (time.perf_counter() - t0) * 1e3
                    warm = []
                    for _ in range(7):
                        t0 = time.perf_counter()
                        call()
                        warm.append((time.perf_counter() - t0) * 1e3)
                    # phases of a warm call
                    t0 = time.perf_counter()
                    lut = _engine.lut_source(model, {})
                    t_lut = (time.perf_counter() - t0) * 1e3
                    ctx = _lib.default_context(local_rank)
                    t0 = time.perf_counter()
                    with ctx.lock:
                        _engine.ensure_luts(ctx, lut, None)
                    t_install = (time.perf_counter() - t0) * 1e3
                    t0 = time.perf_counter()
                    _engine.invert_numpy(lut, None, inc, s_vv, None, None, anc)
                    t_search = (time.perf_counter() - t0) * 1e3
                out[key][f"lut_build_{route}"] = {"first_call_ms": round(first_ms, 2), "warm_call_ms": round(float(np.median(warm)), 3),
                                                  "warm_call_ms_min": round(float(np.min(warm)), 3),
                                                  "warm_phases_ms": {"lut_lookup": round(t_lut, 3), "lut_install_check": round(t_install, 3),
                                                                     "staging_upload_kernels_download_expand": round(t_search, 3)},
                                                  "warm_Mpixels_per_s": round(lines * samples / float(np.median(warm)) / 1e3, 1)}
        out["note"] = ("end to end through the drop-in call: float32 numpy in, complex128 numpy out, sigma0 -> dB by numpy on the host (bit parity); "
                       "first call = fresh libxsw context and no memoised LUT (host route: numpy GMF fill + device interpolation + numpy log10 + "
                       "upload of the 362 MB table; device route: xsw_lut_build), then staging start-up; the reference rebuilds its LUT and "
                       "JIT-compiles its kernel on EVERY call (windspeed.py:144, :306-323)")
    finally:
        options.lut_build, options.device = saved
    return out


def nesz_figures(args, ctx, stream, noise, inc, lines, samples):
    """`nesz_flattening` (windspeed/utils.py:94-163) on resident float32 rasters -> float64: the raster pass in front of the
    dual-pol inversion.  Algorithmic bytes: noise read twice (column means, then the per-line fit) + incidence once + float64
    out = 20 B per pixel.  (The sigma0 raster stands in for a noise raster: same NaN holes, positive values.)"""
    from xsarsea_amd import _lib
    out = torch.empty((lines, samples), dtype=torch.float64, device=noise.device)

    def step():
        ctx.nesz_flatten_raw(lines, samples, _lib.XSW_F32, _lib.MEM_DEVICE, noise.data_ptr(), inc.data_ptr(), out.data_ptr())

    _, ms = time_steps(step, args.steps, 1, stream)  # asynchronous on the stream since round 3 (context-owned scratch)
    dt = ms * 1e-3
    del out
    px = lines * samples
    achieved = 20.0 * px / dt / 1e9
    return {"workload": f"nesz_flattening, {lines}x{samples} float32 noise + incidence -> float64 (k_nesz_colsum + k_nesz_colmean + k_nesz_center + k_nesz_fit + k_nesz_eval)",
            "value": round(px / dt / 1e6, 1), "unit": "Mpixels/s", "ms_per_call": round(dt * 1e3, 3), "bytes_per_pixel": 20,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "note": "HIP events on the launch stream around the call's five launches (asynchronous, context-owned scratch)"}}


# ------------------------------------------------------------------------------------------ rank launcher
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n):
    """`python bench.py --gpus N` without torchrun: start N fresh rank processes (this process has made no GPU call and
    makes none: a process that has initialised the GPU must never exec/fork workers on this pool)."""
    one_dev = os.environ.get("XSW_BENCH_ONE_DEVICE") == "1"
    have = torch.cuda.device_count()  # does not initialise the GPU
    if not one_dev and have < n:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) visible", file=sys.stderr)
        return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    deadline = None
    while procs:
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0 and rc == 0:
                rc = code
                deadline = time.time() + 30.0  # a rank failed: give the others a moment, then stop them
        if deadline is not None and time.time() > deadline:
            for p in procs:
                p.kill()
            deadline = None
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="metric", choices=sorted(CONFIGS),
                    help="BASELINE.json workload: metric = CMOD5.N mono 20000x20000 (default), 2 = 10000x10000 mono, "
                         "3 = dual-pol 20000x20000, 4 = 25000x17000 mono (row-tiled), 5 = CMOD7-shaped LUT 20000x20000")
    ap.add_argument("--lines", type=int, default=None)
    ap.add_argument("--samples", type=int, default=None)
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the configured raster is row-tiled over the ranks (default); "
                         "weak = every rank inverts a full-size tile of an N-times taller raster")
    ap.add_argument("--algo", default="pruned", choices=["pruned", "exhaustive", "exhaustive_f64", "exact"])
    ap.add_argument("--mode", default=None, choices=["mono", "dual", "detrend"],
                    help="override the config's mode; detrend: the sigma0_detrend kernel alone (N = 1 only)")
    ap.add_argument("--resolution", default="high", choices=["high", "low"],
                    help="LUT resolution passed to Model.to_lut: high = the reference default 501x499x181, low = 51x250x73 "
                         "(a parity-test configuration, SURVEY 8d; implies no CPU baseline / exhaustive figure)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the detrend / dB-parity / exhaustive side figures")
    ap.add_argument("--verify-gather", action="store_true",
                    help="N > 1: the ranks' input tiles are gathered on rank 0, which inverts the whole raster in ONE launch and "
                         "compares it bit for bit with the gathered result (exit code 3 on a mismatch)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    # rehearsal knobs (never set by the driver): XSW_BENCH_BACKEND=gloo and XSW_BENCH_ONE_DEVICE=1 let the whole N > 1
    # code path run as several ranks on a one-GPU box (everything but RCCL itself)
    backend = os.environ.get("XSW_BENCH_BACKEND", "nccl")
    if os.environ.get("XSW_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
        if dist.get_world_size() != args.gpus:
            sys.exit(2)
    n_gpus = world
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    cfg = dict(CONFIGS[args.config])
    mode = args.mode or cfg["mode"]
    total_lines = args.lines or cfg["lines"]
    samples = args.samples or cfg["samples"]
    if args.scaling == "weak":
        total_lines *= world

    import xsarsea_amd
    from xsarsea_amd import _lib, multi_gpu
    xsarsea_amd.options.device = local_rank  # one process per GPU: LUT preparation also runs on this rank's device
    timings = {}
    lut, co_dict = build_product_lut(args.resolution, cfg["lut"], timings)
    if args.resolution != "high" or cfg["lut"] != "cmod5n":
        args.no_cpu_baseline = True
    ctx = _lib.Context(local_rank)
    stream = torch.cuda.Stream(device=device)  # the kernels, the events and RCCL all use this stream
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    if cfg["lut"] == "cmod5n":
        # time to a searchable LUT, device route (options.lut_build="device": grid fill + interpolation + dB + layout on the
        # GPU, xsw_lut_build); the timed steps below then run on the host-built table (bit parity with the CPU baseline)
        from xsarsea_amd.windspeed import _engine, get_model
        plan = get_model("gmf_cmod5n").device_lut_plan(**({} if args.resolution == "high" else {"resolution": args.resolution}))
        t0 = time.perf_counter()
        _engine.DeviceLut("gmf_cmod5n", plan[0], plan[1], plan[2], key=None).build(ctx)
        ctx.synchronize()
        timings["lut_device_build_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    t0 = time.perf_counter()
    ctx.upload_luts(co=co_dict)
    ctx.synchronize()
    timings["lut_upload_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    timings["note"] = ("host route = lut_build_ms (Model.to_lut: numpy GMF fill, device interpolation, numpy log10) + lut_upload_ms; "
                       "device route = lut_device_build_ms (xsw_lut_build, table never on the host)")

    l0, l1 = multi_gpu.tile_bounds(total_lines, world, rank)  # this rank's row tile (windspeed.py:356-364: row blocks)
    lines = l1 - l0
    inc, s_vv, anc = make_scene(lines, samples, total_lines, l0, 20260320 + 2 + rank, device)
    torch.cuda.synchronize()
    scene_sum0 = int(s_vv.view(torch.int32).to(torch.int64).sum().item())  # (--verify-gather checks that nothing wrote into the inputs since)
    if mode == "detrend":
        if rank == 0:
            d = detrend_figures(args, ctx, stream, s_vv, lines, samples)
            print(json.dumps({
                "metric": "Mpixels/s sigma0_detrend", "value": d["value"], "unit": "Mpixels/s", "n_gpus": 1, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": d["kernel_ms"], "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": d["workload"]},
                "roofline": dict(d["roofline"], kernel="k_detrend", kernel_ms=d["kernel_ms"], bytes_per_pixel=12)}))
        return
    out = torch.empty((lines, samples), dtype=torch.complex64, device=device)
    s_vh = dsig = out_dual = None
    if mode == "dual":
        from xsarsea_amd.windspeed import _engine, get_model
        ctx.upload_luts(cr=_engine._cr_dict(get_model("gmf_s1_v2")._lut(units="dB")))
        s_vh, dsig = make_crosspol(inc, anc, 777 + rank, device)
        out_dual = torch.empty((lines, samples), dtype=torch.complex64, device=device)
    algo = _lib.ALGOS[args.algo]

    # N > 1: the library's own gathered tiling, `multi_gpu.TiledPipeline` through `multi_gpu.invert_tiled_device` -- what
    # `multi_gpu.invert_from_model_tiled(gather=True)` runs for a user: the tile is inverted in row chunks to 4-byte GRID CODES
    # (a quarter of the complex64 bytes; half for dual-pol), chunk k travels to rank 0 over xGMI while chunk k + 1 is inverted,
    # rank 0 expands every chunk on a side stream as it lands.  Nothing of that choreography lives in this file.
    coded = world > 1
    n_chunks = 8 if world > 1 else 1
    pipe = multi_gpu.TiledPipeline(total_lines, samples, dual=(mode == "dual"), device=device, dst=0, n_chunks=n_chunks,
                                   out_dtype=torch.complex64) if coded else None

    def invert_rows(r0, r1, s0_ptr=None, is_db=False):
        off = r0 * samples
        co_ptr = (s0_ptr if s0_ptr is not None else s_vv.data_ptr()) + off * 4
        if mode == "dual":
            ctx.invert_raw(r1 - r0, samples, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, inc.data_ptr() + off * 4, co_ptr,
                           s_vh.data_ptr() + off * 4, dsig.data_ptr() + off * 4, anc.data_ptr() + off * 8,
                           out.data_ptr() + off * 8, out_dual.data_ptr() + off * 8, algo=algo, dual_select=True, sigma0_is_db=is_db)
        else:
            ctx.invert_raw(r1 - r0, samples, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, inc.data_ptr() + off * 4, co_ptr,
                           None, None, anc.data_ptr() + off * 8, out.data_ptr() + off * 8, None, algo=algo, sigma0_is_db=is_db)

    def step(gathering=True):
        if coded and gathering:
            multi_gpu.invert_tiled_device(ctx, inc, s_vv, anc, total_lines, sigma0_cr=s_vh, dsig_cr=dsig, pipeline=pipe, algo=algo,
                                          dual_select=True, wait=False)
            return
        for k in range(n_chunks):  # the tiles left where they are computed (complex64): no exchange
            r0, r1 = multi_gpu.chunk_bounds(lines, n_chunks, k)
            if r1 > r0:
                invert_rows(r0, r1)

    def gather():
        if coded:
            pipe.finish()  # senders: their sends; rank 0: the launch stream continues behind the last chunk's expansion

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:  # connect every peer to rank 0 once (RCCL builds its point-to-point channels lazily), whatever --warmup is
        hello = torch.zeros((world, samples), dtype=torch.complex64, device=device)
        for q in multi_gpu.gather_chunk_async(hello[rank:rank + 1] if rank else hello[:1], world, 0, 1, dst=0, out=hello):
            q.wait()
        del hello
    for _ in range(args.warmup):
        step()
        gather()
    fence()
    ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(args.steps)]
    ctx.timing_enable(True)  # HIP events of the library around each of the two kernels of an inversion (launch stream)
    t0 = time.perf_counter()
    for a, b, c_ in ev:
        a.record(stream)
        step()
        b.record(stream)
        gather()
        c_.record(stream)  # N > 1: after the last transfer has landed and the codes are expanded
    fence()
    dt = time.perf_counter() - t0
    step_kernels_ms = float(np.mean([a.elapsed_time(b) for a, b, _ in ev]))  # all kernels of a step on this rank
    step_total_ms = float(np.mean([a.elapsed_time(c_) for a, _, c_ in ev]))   # + waits for the transfers + expansion
    tm = ctx.timing()
    ctx.timing_enable(False)
    if tm["launches"]:  # two-kernel path: the dominant kernel is k_invert_band; per STEP = summed over the step's row chunks
        kernel_ms = tm["first_kernel_ms"] / args.steps
        second_ms = (tm["second_kernel_ms"] + tm.get("band2_kernel_ms", 0.0) + tm.get("blocks_kernel_ms", 0.0)) / args.steps  # k_invert_band2 + k_invert_blocks + k_invert_list
    else:
        kernel_ms, second_ms = step_kernels_ms, None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    mg = None
    if world > 1:
        # what explains a scaling curve: every rank's kernel time, the bytes rank 0 ingests, the raw transfer time of those
        # bytes (a gather of the already computed codes, nothing else in flight), the part of the step the transfers and the
        # expansion are NOT hidden behind the kernels, and the same steps with the outputs left sharded (no exchange at all)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "lines": lines, "kernel_ms": round(kernel_ms + (second_ms or 0.0), 3)})
        fence()
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tg = time.perf_counter()
        g0.record(stream)
        for _ in range(args.steps):
            pipe.gather_only()
        g1.record(stream)
        fence()
        gather_only_ms = (time.perf_counter() - tg) / args.steps * 1e3
        for _ in range(max(args.warmup, 1)):
            step(gathering=False)
        fence()
        tn = time.perf_counter()
        for _ in range(args.steps):
            step(gathering=False)
        fence()
        t = torch.tensor([time.perf_counter() - tn], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_ng = float(t.item())
        code_bytes = 4 if mode == "mono" else 8
        gbytes = multi_gpu.gather_bytes_into(total_lines, samples, world, 0, code_bytes)
        mg = {"gather": "4-byte grid codes per pixel and search (xsw_invert out_code_*), expanded to complex64 on rank 0 (xsw_expand_codes) inside the step",
              "gather_bytes": int(gbytes), "gather_bytes_if_complex64": int(gbytes * 2),
              "gather_only_ms": round(gather_only_ms, 3), "gather_only_GBps": round(gbytes / (gather_only_ms * 1e-3) / 1e9, 1),
              "step_total_ms_rank0": round(step_total_ms, 3), "step_kernels_ms_rank0": round(step_kernels_ms, 3),
              "exposed_gather_ms": round(step_total_ms - step_kernels_ms, 3),
              "kernel_ms_per_rank": per_rank,
              "no_gather": {"value": round(total_lines * samples * args.steps / dt_ng / 1e6, 3), "unit": "Mpixels/s",
                            "ms_per_step": round(dt_ng / args.steps * 1e3, 3),
                            "note": "the same steps with complex64 tiles left on their ranks (what a dask consumer of row blocks does): no exchange"},
              "note": "gather_only_ms: the gather alone (codes already computed, nothing else in flight), wall clock between barriers; "
                      "exposed_gather_ms: rank 0's step from first launch to the expanded raster minus its kernels' time"}

    # evaluated-work counters: one extra, untimed pass with the device-side statistics on
    ctx.stats_enable(True)
    step()
    gather()
    fence()
    stats = ctx.stats()
    ctx.stats_enable(False)
    if world > 1:
        t = torch.tensor([stats["pixels_co"], stats["cand_co"], stats["pixels_exact"]], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        stats["pixels_co"], stats["cand_co"], stats["pixels_exact"] = (int(x) for x in t.tolist())
        t = torch.tensor([kernel_ms], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kernel_ms_max = float(t.item())
    else:
        kernel_ms_max = kernel_ms

    gather_ok = None
    w_inc = w_s = w_anc = w_vh = w_dsig = None
    if args.verify_gather and world > 1:
        # the rasters the ranks have ACTUALLY been inverting, gathered on rank 0 (every rank takes part): the comparison below then checks
        # the pipeline and nothing else -- a regenerated scene would also have to trust the generator to reproduce itself call to call
        def gathered(t):
            full_t = torch.empty((total_lines, samples), dtype=t.dtype, device=device) if rank == 0 else None
            if not (backend == "nccl"):
                torch.cuda.synchronize()
            for q in multi_gpu.gather_chunk_async(t, total_lines, 0, 1, dst=0, out=full_t):
                q.wait()
            torch.cuda.synchronize()
            return full_t
        w_inc, w_s, w_anc = gathered(inc), gathered(s_vv), gathered(anc)
        if mode == "dual":
            w_vh, w_dsig = gathered(s_vh), gathered(dsig)
    if args.verify_gather and world > 1 and rank == 0:
        w_dual = torch.empty((total_lines, samples), dtype=torch.complex64, device=device) if mode == "dual" else None
        # diagnostic: does the generator reproduce rank 0's tile?  (bit for bit, same seed; a difference is the generator's, not the pipeline's)
        again = make_scene(l1 - l0, samples, total_lines, l0, 20260320 + 2, device)
        torch.cuda.synchronize()
        if not bool(torch.equal(again[1].view(torch.int32), s_vv.view(torch.int32))):
            rr = (again[1].view(torch.int32) != s_vv.view(torch.int32)).any(dim=1).nonzero().flatten()
            print(f"bench.py: --verify-gather: note: the scene generator did not reproduce rank 0's tile bit for bit (sigma0 checksum at start {scene_sum0}, now "
                  f"{int(s_vv.view(torch.int32).to(torch.int64).sum().item())}; lines {rr[:3].tolist()}..{rr[-2:].tolist()}); the check below uses the gathered inputs",
                  file=sys.stderr)
        del again
        w_out = torch.empty((total_lines, samples), dtype=torch.complex64, device=device)
        ctx.invert_raw(total_lines, samples, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, w_inc.data_ptr(), w_s.data_ptr(),
                       w_vh.data_ptr() if mode == "dual" else None, w_dsig.data_ptr() if mode == "dual" else None,
                       w_anc.data_ptr(), w_out.data_ptr(), w_dual.data_ptr() if mode == "dual" else None, algo=algo,
                       dual_select=(mode == "dual"))
        torch.cuda.synchronize()
        bits = lambda t: torch.view_as_real(t).view(torch.int32)
        gather_ok = bool(torch.equal(bits(w_out), bits(pipe.full)))
        if mode == "dual":
            gather_ok = gather_ok and bool(torch.equal(bits(w_dual), bits(pipe.full_dual)))
        if not gather_ok:  # where: which rank's tile, which chunk of it
            bad = (bits(w_out) != bits(pipe.full)).any(dim=-1)
            rows = bad.any(dim=1).nonzero().flatten()
            print(f"bench.py: --verify-gather: {int(bad.sum().item())} pixels differ in {int(rows.numel())} lines; first lines {rows[:8].tolist()}, "
                  f"last {rows[-3:].tolist()}; tiles {[multi_gpu.tile_bounds(total_lines, world, r) for r in range(world)]}", file=sys.stderr)
            ij = bad.nonzero()[:4].tolist()
            print("bench.py: --verify-gather: (line, sample, one launch, gathered, code): " +
                  "; ".join(f"({l}, {c}, {complex(w_out[l, c].item()):.4f}, {complex(pipe.full[l, c].item()):.4f}, 0x{int(pipe.full_codes[l, c].item()) & 0xffffffff:08x})"
                            for l, c in ij), file=sys.stderr)
            ctx.expand_codes_on_stream(stream.cuda_stream, total_lines * samples, _lib.XSW_F32, pipe.full_codes.data_ptr(), None, w_out.data_ptr(), None)
            torch.cuda.synchronize()  # (w_out now = the gathered CODES expanded once more: equal to pipe.full unless the expansion raced)
            print(f"bench.py: --verify-gather: the gathered codes expanded once more differ from the gathered winds in "
                  f"{int((bits(w_out) != bits(pipe.full)).any(dim=-1).sum().item())} pixels", file=sys.stderr)
        del w_inc, w_s, w_anc, w_out, w_vh, w_dsig, w_dual

    if rank == 0:
        px_total = total_lines * samples
        value = px_total * args.steps / dt / 1e6
        bytes_px = (BYTES_READ_PX + BYTES_WRITE_PX) if mode == "mono" else (24 + 16)  # dual: +vh, +dsig; 2 outputs
        if coded:  # N > 1: the kernels write 4-byte codes instead of complex64
            bytes_px = (BYTES_READ_PX + 4) if mode == "mono" else (24 + 8)
        # the dominant kernel reads every pixel's inputs and writes the pixels it decides itself (the ones it hands to
        # k_invert_band2 / k_invert_list are written there); the chain as a whole moves bytes_px per pixel
        read_px = BYTES_READ_PX if mode == "mono" else 24
        handed = (tm.get("last_band2_pixels", 0) + tm.get("last_blocks_pixels", 0) + tm.get("last_list_pixels", 0)) if second_ms is not None else 0
        achieved = (read_px * lines * samples + (bytes_px - read_px) * (lines * samples - handed)) / (kernel_ms * 1e-3) / 1e9  # rank 0's tile / rank 0's kernel time
        chain_ms = kernel_ms + (second_ms or 0.0)
        chain_ms_max = kernel_ms_max + (second_ms or 0.0)
        is_metric_shape = (mode == "mono" and args.resolution == "high" and cfg["lut"] == "cmod5n" and args.algo == "pruned"
                           and (lines, samples) == (20000, 20000))
        traffic, traffic_prov, chain_traffic = None, None, None
        if is_metric_shape:
            tj, traffic_prov = fresh_profile(f"{PROFILE_ROUND}_hbm_traffic_summary.json")
            if tj:
                # the dominant kernel's own traffic when the profile has it per kernel (else the whole chain's)
                per = tj.get("per_kernel_20000x20000", {}).get("k_invert_band")
                traffic = int((per or tj["k_invert_20000x20000"])["hbm_bytes_per_launch"])
                chain_traffic = int(tj["k_invert_20000x20000"]["hbm_bytes_per_launch"])
        cand_full = int(lut.shape[1] * lut.shape[2])
        evaluated = stats["cand_co"] / max(stats["pixels_co"], 1)
        lane_ops = OPS_PER_CANDIDATE * stats["cand_co"] / world / (chain_ms_max * 1e-3) if args.algo == "pruned" else \
            OPS_PER_CANDIDATE * cand_full * lines * samples / (kernel_ms * 1e-3)
        valu = {"bound": "valu", "unit": "lane-ops/s", "peak": LANE_OPS_PEAK,
                "ops_per_candidate": OPS_PER_CANDIDATE,
                "candidates_per_pixel_full_grid": cand_full,
                "evaluated_candidates_per_pixel": round(evaluated, 1),
                "pixels_exact_fallback": stats["pixels_exact"],
                "achieved": float(f"{lane_ops:.4g}"), "frac": round(lane_ops / LANE_OPS_PEAK, 5),
                "peak_packed_fp32": LANE_OPS_PEAK, "peak_issue": LANE_OPS_PEAK_ISSUE, "frac_of_issue_peak": round(lane_ops / LANE_OPS_PEAK_ISSUE, 5),
                "note": "useful work only: 6 lane-ops x candidates scored (statistics pass, every window swept in k_invert_band) / the chain's kernel time. "
                        "`frac` is against `peak` = peak_packed_fp32 (256 CU x 128 lanes x 2.4 GHz, SURVEY 8d's figure: packed FP32); `frac_of_issue_peak` against "
                        "peak_issue (64 lanes per CU per clock: one wave64 instruction per 4 cycles per SIMD), the peak the counter figures under `issue` "
                        "(valu_issue_frac_of_simd_cycles) are measured against -- the two peaks differ by 2; "
                        "the grid has candidates_per_pixel_full_grid points, all but the evaluated ones are excluded by an exact bound"}
        sq, sq_prov = (fresh_profile(f"{PROFILE_ROUND}_pmc_counters_summary.json") if is_metric_shape else (None, None))
        sq = (sq or {}).get("k_invert_band")
        if sq_prov and "stale_profile" in sq_prov:
            valu["issue"] = sq_prov
        if sq and is_metric_shape:
            try:
                pp = sq["per_pixel"]
                cyc = sq["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
                valu["issue"] = {"kernel": "k_invert_band",
                                 "valu_insts_per_pixel": round(pp["SQ_INSTS_VALU"], 1), "salu_insts_per_pixel": round(pp["SQ_INSTS_SALU"], 1),
                                 "vmem_rd_insts_per_pixel": round(pp["SQ_INSTS_VMEM_RD"], 1),
                                 "l1_line_accesses_per_pixel": round(pp["TCP_TOTAL_CACHE_ACCESSES_sum"], 1),
                                 "valu_issue_frac_of_simd_cycles": round(sq["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cyc), 3),
                                 "texture_addresser_busy_frac": round(sq["TA_TA_BUSY_sum"] / (256.0 * cyc), 3),
                                 "wave_time_waiting_frac": round(sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], 3),
                                 "valu_int32_insts_per_pixel": round(pp.get("SQ_INSTS_VALU_INT32", float("nan")), 1),
                                 "branch_insts_per_pixel": round(pp.get("SQ_INSTS_BRANCH", float("nan")), 1),
                                 "measured_on": sq_prov.get("measured_on"),
                                 "source": f"profiles/{PROFILE_ROUND}_pmc_counters_summary.json (rocprofv3 --pmc passes of profiles/collect_counters.sh, "
                                           "same workload; 4 issue cycles per wave64 VALU instruction, 1024 SIMDs, 256 TAs, cycles = GRBM_GUI_ACTIVE / 8 XCDs)"}
            except Exception:
                pass
        mode_txt = "mono-VV" if mode == "mono" else "dual-pol (VV + S1 VH GMF)"
        par = f"row tiles x{n_gpus}" + (f" ({args.scaling} scaling: {'the same raster split' if args.scaling == 'strong' else 'one full tile per rank'}), "
                                        f"RCCL gather of 4-byte grid codes to rank 0 in the step ({n_chunks} chunks behind the kernel), expanded to complex64 there"
                                        if n_gpus > 1 else "")
        res = {
            "metric": "Mpixels/s wind inversion (CMOD5.N, 20k x 20k sigma0)",
            "value": round(value, 3), "unit": "Mpixels/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{'CMOD7-shaped table' if cfg['lut'] == 'cmod7' else 'CMOD5.N'} {mode_txt} inversion ({cfg['note']}), "
                                   f"{total_lines}x{samples} float32 sigma0/incidence + complex64 ancillary, "
                                   f"{'default' if args.resolution == 'high' else 'resolution=low'} LUT "
                                   f"{'x'.join(str(int(x)) for x in lut.shape)} ({cand_full} candidates/pixel), "
                                   f"complex64 out, algo={args.algo}, sigma0 -> dB fused on the device",
                       "baseline_config": args.config, "lines": total_lines, "samples": samples,
                       "lines_rank0": lines, "lut": [int(x) for x in lut.shape], "parallelism": par},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_provenance": traffic_prov,
                         "kernel": {"exhaustive": "k_invert_exhaustive32", "exhaustive_f64": "k_invert_exhaustive"}.get(
                             args.algo, "k_invert_band" if second_ms is not None else "k_invert"),
                         "kernel_ms": round(kernel_ms, 3), "bytes_per_pixel": bytes_px,
                         "second_kernel": None if second_ms is None else dict(
                             {"kernel": "k_invert_list", "kernel_ms": round(second_ms, 3), "pixels_last_launch": tm.get("last_list_pixels")},
                             **({"kernel": "k_invert_band2 + k_invert_blocks + k_invert_list", "k_invert_band2_ms": round(tm.get("band2_kernel_ms", 0.0) / args.steps, 3),
                                 "k_invert_blocks_ms": round(tm.get("blocks_kernel_ms", 0.0) / args.steps, 3),
                                 "k_invert_list_ms": round(tm["second_kernel_ms"] / args.steps, 3),
                                 "pixels_to_band2_last_launch": tm.get("last_band2_pixels"),
                                 "pixels_to_blocks_last_launch": tm.get("last_blocks_pixels")} if tm.get("band2_kernel_ms", 0.0) > 0.0 else {})),
                         "chain": {"kernels": "k_invert_band + k_invert_band2 + k_invert_blocks + k_invert_list" if second_ms is not None else "k_invert",
                                   "ms": round(chain_ms, 3), "achieved": round(bytes_px * lines * samples / (chain_ms * 1e-3) / 1e9, 3),
                                   "frac": round(bytes_px * lines * samples / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                                   "traffic": chain_traffic},
                         "step_kernels_ms": round(step_kernels_ms, 3),
                         "note": f"algorithmic raster bytes ({read_px} B read per pixel x rank 0's {lines * samples} px + {bytes_px - read_px} B written per "
                                 "pixel the kernel decides itself, i.e. not handed to k_invert_band2 / k_invert_blocks / k_invert_list) / mean "
                                 "duration of the dominant kernel per step (HIP events on the launch stream, recorded by the library around each kernel); "
                                 "chain: all bytes of the step / the four kernels' time; "
                                 "the search itself is bound by VALU issue and the texture-address path, not by HBM: see valu.  "
                                 "north_star's '>= 40 % of the HBM-read roofline' is OUT OF REACH for an exact search: at 40 % of 8 TB/s a pixel's 16 B of "
                                 "input may cost 5 ps, i.e. ~5 wave-instructions (300 lane-instructions) end to end, while the exact branch-and-bound "
                                 "spends 44 VALU + 18 SALU wave-instructions per pixel (~2 800 lane-instructions: 16 for stage 1 -- dB, incidence bin, three "
                                 "rays, window --, 25 for the band passes, 3 for the store) at 86 % VALU issue: the binding roofline is VALU issue, "
                                 "and the kernel sits at 3.5 % of HBM (2.3 % counting reads only) because it is 12x over that instruction budget, not because it wastes bytes "
                                 "(counter traffic 1.1x the algorithmic bytes)",
                         "valu": valu},
            "lut": timings,
        }
        extras = n_gpus == 1 and not args.no_extras
        if extras and mode == "mono":
            # the bit-parity configuration for float32 rasters: sigma0 converted to dB by numpy on the host (as the
            # drop-in API does by default, options.db_on_device="auto") and resident in HBM before the timed region
            from xsarsea_amd.windspeed import _engine
            t0 = time.perf_counter()
            s_db = torch.from_numpy(_engine._to_db(s_vv.cpu().numpy())).to(device)
            host_db_s = time.perf_counter() - t0
            torch.cuda.synchronize()
            dbt, db_ms = time_steps(lambda: invert_rows(0, lines, s_db.data_ptr(), True), args.steps, 1, stream)
            res["parity_config"] = {"db_mode": "host numpy float32 log10 (bit-parity with the reference CPU path), sigma0_is_db=1",
                                    "value": round(px_total * args.steps / dbt / 1e6, 3), "unit": "Mpixels/s",
                                    "kernel_ms": round(db_ms, 3), "host_db_conversion_s_untimed": round(host_db_s, 2),
                                    "note": "kernel only, dB raster resident; parity_config_end_to_end times everything"}
            del s_db
            res["parity_config_end_to_end"] = end_to_end_parity_figure(ctx, _lib, inc, s_vv, anc, out, lines, samples, algo)
            res["host_path"] = host_path_figure(inc, s_vv, anc, samples)
        if extras and mode == "mono" and args.algo == "pruned" and not args.no_cpu_baseline:
            # like-for-like figure: the literal exhaustive sweep (every one of the 90319 candidates scored per pixel,
            # LUT tiled through LDS, float32 screening + float64 settle) on the first lines of the same raster
            xl = max(4, min(lines, 2000))
            ex_alg = _lib.ALGOS["exhaustive"]

            def xstep():
                ctx.invert_raw(xl, samples, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, inc.data_ptr(), s_vv.data_ptr(),
                               None, None, anc.data_ptr(), out.data_ptr(), None, algo=ex_alg)
            _, xms = time_steps(xstep, 1, 1, stream)
            xops = OPS_PER_CANDIDATE * cand_full * xl * samples / (xms * 1e-3)
            res["exhaustive"] = {"value": round(xl * samples / (xms * 1e-3) / 1e6, 3), "unit": "Mpixels/s",
                                 "workload": f"first {xl} lines x {samples} samples of the same raster, every candidate scored",
                                 "kernel": "k_invert_exhaustive32", "kernel_ms": round(xms, 3),
                                 "valu_frac": round(xops / LANE_OPS_PEAK, 4)}
        if extras and mode == "mono" and args.algo == "pruned" and cfg["lut"] == "cmod5n" and args.resolution == "high":
            res["hard_scene"] = hard_scene_figures(ctx, _lib, stream, device, samples, algo, min(4000, lines))
        if extras:
            res["gmf_eval"] = gmf_eval_figures(args, ctx, stream, device)
            res["latency"] = latency_figures(local_rank)
            res["detrend"] = detrend_figures(args, ctx, stream, s_vv, lines, samples)
            res["nesz_flatten"] = nesz_figures(args, ctx, stream, s_vv, inc, lines, samples)
        if n_gpus == 1 and not args.no_cpu_baseline and mode == "mono":
            cpu, parity = cpu_baseline_and_parity(ctx, inc, s_vv, anc, args.algo)
            res["cpu_baseline"] = cpu
            parity["db_mode"] = ("timed value: sigma0 -> dB fused on the device (float32 log10 correctly rounded; index_match_device_db "
                                 "of pixels agree with numpy's few-ulp float32 log10, the rest differ by one grid step); "
                                 "parity_config: host-converted dB, index_match_host_db")
            res["parity"] = parity
        if mg is not None:
            res["multi_gpu"] = mg
            # the same job with the outputs left where they are computed (what a dask consumer of row blocks does; the API's
            # invert_from_model_tiled(..., gather=False)): no exchange -- the second top-level figure of an N > 1 line
            res["value_no_gather"] = dict(mg["no_gather"], n_gpus=n_gpus)
            if mode == "mono" and not args.no_extras:
                res["host_path_all_devices"] = host_path_all_devices_figure(inc, s_vv, anc, samples, n_gpus)
                torch.cuda.set_device(local_rank)  # (creating contexts on the other GPUs moved this thread's current device)
            res["multi_gpu"]["hardware_note"] = ("first hardware numbers of the N > 1 path come from the driver's scaling run: no multi-GPU node was available to the build "
                                                 "(rehearsed with gloo on one device)")
        if gather_ok is not None:
            res["gather_verified"] = gather_ok
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if gather_ok is False:
        sys.exit(3)


if __name__ == "__main__":
    main()
