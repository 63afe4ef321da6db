"""GPU: grid codes (xsw_invert_args.out_code_*, xsw_expand_codes), the host-memory pipeline that ships them over PCIe,
page-locked rasters, the capped work list (overflow and allocation-failure routes) and stream switching.

The rule throughout: whatever route a pixel takes, the bits are those the device-raster path stores to out_co / out_cr
(which the golden / oracle tests of test_gpu_kernel.py pin to the reference)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from test_gpu_kernel import synthetic_scene
from util import lut_dicts

pytestmark = pytest.mark.gpu


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32 if a.dtype == np.complex64 else np.uint64)


def _device_run(ctx, torch, _lib, arrs, out_c, want, dual_select=False, is_db=False, algo=None):
    """xsw_invert on device rasters; want = subset of {"complex", "codes"}.  Returns dict of host arrays."""
    dev = torch.device("cuda", 0)
    inc, s_co, s_cr, dsig, anc = arrs
    t = {k: (None if v is None else torch.from_numpy(np.ascontiguousarray(v)).to(dev))
         for k, v in dict(inc=inc, s_co=s_co, s_cr=s_cr, dsig=dsig, anc=anc).items()}
    shape = inc.shape
    cdt = torch.complex64 if out_c == np.complex64 else torch.complex128
    o = {}
    if "complex" in want:
        o["co"] = torch.empty(shape, dtype=cdt, device=dev) if s_co is not None else None
        o["cr"] = torch.empty(shape, dtype=cdt, device=dev) if s_cr is not None else None
    if "codes" in want:
        o["cc"] = torch.full(shape, 0x12345678, dtype=torch.int32, device=dev) if s_co is not None else None
        o["ccr"] = torch.full(shape, 0x12345678, dtype=torch.int32, device=dev) if s_cr is not None else None
    torch.cuda.synchronize()
    p = lambda x: None if x is None else x.data_ptr()
    f32 = inc.dtype == np.float32
    ctx.invert_raw(shape[0], shape[1], _lib.XSW_F32 if f32 else _lib.XSW_F64, _lib.XSW_F32 if out_c == np.complex64 else _lib.XSW_F64,
                   _lib.MEM_DEVICE, p(t["inc"]), p(t["s_co"]), p(t["s_cr"]), p(t["dsig"]), p(t["anc"]), p(o.get("co")), p(o.get("cr")),
                   None, dsig_cr_scalar=0.1, sigma0_is_db=is_db, algo=_lib.ALGO_PRUNED if algo is None else algo, dual_select=dual_select,
                   out_code_co=p(o.get("cc")), out_code_cr=p(o.get("ccr")))
    ctx.synchronize()
    return o


@pytest.mark.parametrize("out_c", [np.complex64, np.complex128])
@pytest.mark.parametrize("mode", ["mono", "dual", "dual_select", "cross_only"])
def test_codes_expand_to_the_stored_winds(gpu_ctx, default_luts, mode, out_c):
    """Device rasters: complex outputs and grid codes from ONE launch; the codes expanded on the device (k_expand) and on
    the host (expand_host) give the complex outputs bit for bit, NaN kinds included."""
    torch = pytest.importorskip("torch")
    from xsarsea_amd import _lib
    lco, lcr = default_luts
    co, cr = lut_dicts(lco, lcr)
    gpu_ctx.upload_luts(co=co, cr=cr)
    inc, s_vv, s_vh, dsig, anc = synthetic_scene(70, 333, np.float32, 11)
    anc[5, 40:60] = np.nan          # sigma0 valid but ancillary NaN -> (nan, 0)
    s_vh[9, 100:130] = np.nan       # no cross-pol search -> (nan, nan) / co-pol when selected
    arrs = dict(mono=(inc, s_vv, None, None, anc), dual=(inc, s_vv, s_vh, dsig, anc), dual_select=(inc, s_vv, s_vh, dsig, anc),
                cross_only=(inc, None, s_vh, None, None))[mode]
    o = _device_run(gpu_ctx, torch, _lib, arrs, out_c, {"complex", "codes"}, dual_select=(mode == "dual_select"))
    dev = torch.device("cuda", 0)
    n = inc.size
    od = _lib.XSW_F32 if out_c == np.complex64 else _lib.XSW_F64
    cdt = torch.complex64 if out_c == np.complex64 else torch.complex128
    p = lambda x: None if x is None else x.data_ptr()
    # device expansion
    e_co = torch.empty(inc.shape, dtype=cdt, device=dev) if o["cc"] is not None else None
    e_cr = torch.empty(inc.shape, dtype=cdt, device=dev) if o["ccr"] is not None else None
    gpu_ctx.expand_codes_raw(n, _lib.MEM_DEVICE, od, p(o["cc"]), p(o["ccr"]), p(e_co), p(e_cr))
    gpu_ctx.synchronize()
    # host expansion
    h = {k: (None if v is None else v.cpu().numpy()) for k, v in o.items()}
    hc = {k: (None if h[k] is None else h[k].view(np.uint32)) for k in ("cc", "ccr")}
    x_co = np.empty(inc.shape, out_c) if hc["cc"] is not None else None
    x_cr = np.empty(inc.shape, out_c) if hc["ccr"] is not None else None
    hp = lambda a: None if a is None else a.ctypes.data
    gpu_ctx.expand_codes_raw(n, _lib.MEM_HOST, od, hp(hc["cc"]), hp(hc["ccr"]), hp(x_co), hp(x_cr))
    for name, ref, dv, hv in (("co", h["co"], e_co, x_co), ("cr", h["cr"], e_cr, x_cr)):
        if ref is None:
            continue
        assert np.array_equal(_bits(ref), _bits(dv.cpu().numpy())), f"{mode}/{name}: device expansion differs from the stored winds"
        assert np.array_equal(_bits(ref), _bits(hv)), f"{mode}/{name}: host expansion differs from the stored winds"
    if hc["cc"] is not None:
        assert not np.any(hc["cc"] == 0x12345678), "a co-pol code was not written"
        assert np.any(hc["cc"] == _lib.CODE_NAN_RE) and np.all(np.isnan(h["co"].real) == (hc["cc"] >= _lib.CODE_NAN))
    if hc["ccr"] is not None:
        assert not np.any(hc["ccr"] == 0x12345678), "a cross-pol code was not written"
        if mode == "dual_select":
            assert np.any((hc["ccr"] & _lib.CODE_PICK_CO) != 0) and np.any((hc["ccr"] & _lib.CODE_PICK_CO) == 0)


@pytest.mark.parametrize("dtype,out_c", [(np.float32, np.complex128), (np.float32, np.complex64), (np.float64, np.complex128)])
def test_host_pipeline_equals_device_rasters(gpu_ctx, default_luts, dtype, out_c):
    """Host rasters (page-locked staging ring, worker threads, codes over PCIe, host expansion; several chunks, ragged last
    chunk) == the same rasters inverted in HBM, bit for bit: winds, codes and the out_idx triple; also with page-locked
    caller rasters (XSW_MEM_HOST_PINNED) and with 1 and 3 worker threads."""
    torch = pytest.importorskip("torch")
    from xsarsea_amd import _lib
    lco, lcr = default_luts
    co, cr = lut_dicts(lco, lcr)
    gpu_ctx.upload_luts(co=co, cr=cr)
    inc, s_vv, s_vh, dsig, anc = synthetic_scene(531, 700, dtype, 23)  # 371 700 px: 6 chunks of 96 lines, the last one ragged
    ref = _device_run(gpu_ctx, torch, _lib, (inc, s_vv, s_vh, dsig, anc), out_c, {"complex", "codes"}, dual_select=True)
    ref = {k: v.cpu().numpy() for k, v in ref.items()}
    try:
        for threads, pinned in ((0, False), (1, False), (3, False), (0, True)):
            gpu_ctx.set_host_threads(threads)
            a = (inc, s_vv, s_vh, dsig, anc)
            if pinned:
                a = []
                for x in (inc, s_vv, s_vh, dsig, anc):
                    pa = gpu_ctx.pinned_empty(x.shape, x.dtype)
                    pa[...] = x
                    a.append(pa)
            g = gpu_ctx.invert_host(a[0], sigma0_co=a[1], sigma0_cr=a[2], dsig_cr=a[3], anc=a[4], dual_select=True, out_dtype=out_c,
                                    want_idx=True, want_codes=True, algo="pruned", pinned=pinned)
            what = f"threads={threads} pinned={pinned}"
            assert np.array_equal(_bits(g[0]), _bits(ref["co"])), what
            assert np.array_equal(_bits(g[1]), _bits(ref["cr"])), what
            assert np.array_equal(g[3][0], ref["cc"].view(np.uint32)) and np.array_equal(g[3][1], ref["ccr"].view(np.uint32)), what
            # out_idx as the device-raster path reports it: from the codes
            cc, ccr = g[3]
            have = cc < _lib.CODE_NAN
            flat = (cc & 0x3FFFFFFF).astype(np.int64)
            assert np.array_equal(g[2][..., 0], np.where(have, flat // len(lco.phi), -1))
            assert np.array_equal(g[2][..., 1], np.where(have, flat % len(lco.phi), -1))
            icr = (ccr & _lib.CODE_NO_INDEX).astype(np.int64)
            assert np.array_equal(g[2][..., 2], np.where((ccr == _lib.CODE_NAN_RE) | (icr == _lib.CODE_NO_INDEX), -1, icr))
            if pinned:
                for pa in a:
                    gpu_ctx.pinned_free(pa)
    finally:
        gpu_ctx.set_host_threads(0)


def test_detrend_and_nesz_host_pipelines(gpu_ctx):
    """The worker pipeline behind xsw_detrend / xsw_nesz_flatten on host rasters == the device-raster kernels, bit for bit
    (several chunks; 1 and default worker threads)."""
    torch = pytest.importorskip("torch")
    from xsarsea_amd import _lib
    rng = np.random.default_rng(3)
    lines, samples = 1111, 1000
    sig = rng.gamma(2.0, 0.01, (lines, samples)).astype(np.float32)
    sig[rng.random(sig.shape) < 0.01] = np.nan
    ratio = rng.uniform(0.5, 2.0, samples)
    inc = (30 + 15 * np.arange(samples) / samples + rng.normal(0, 1e-3, (lines, samples))).astype(np.float32)
    dev = torch.device("cuda", 0)
    t_sig, t_inc = torch.from_numpy(sig).to(dev), torch.from_numpy(inc).to(dev)
    d_det = torch.empty((lines, samples), dtype=torch.float64, device=dev)
    d_nz = torch.empty((lines, samples), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    gpu_ctx.detrend_raw(lines, samples, _lib.XSW_F32, _lib.XSW_F64, _lib.MEM_DEVICE, t_sig.data_ptr(), ratio, d_det.data_ptr())
    gpu_ctx.nesz_flatten_raw(lines, samples, _lib.XSW_F32, _lib.MEM_DEVICE, t_sig.data_ptr(), t_inc.data_ptr(), d_nz.data_ptr())
    gpu_ctx.nesz_flatten_raw(lines, samples, _lib.XSW_F32, _lib.MEM_DEVICE, t_sig.data_ptr(), t_inc.data_ptr(), d_nz.data_ptr())  # stream-ordered reuse of the scratch
    gpu_ctx.synchronize()
    try:
        for threads in (1, 0):
            gpu_ctx.set_host_threads(threads)
            h_det = gpu_ctx.detrend_host(sig, ratio)
            h_nz = gpu_ctx.nesz_flatten_host(sig, inc)
            assert np.array_equal(h_det.view(np.uint64), d_det.cpu().numpy().view(np.uint64))
            assert np.array_equal(h_nz.view(np.uint64), d_nz.cpu().numpy().view(np.uint64))
    finally:
        gpu_ctx.set_host_threads(0)


_OVERFLOW_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {repo!r}); sys.path.insert(0, {repo!r} + "/tests")
from oracle import lut as olut
from util import lut_dicts
from test_gpu_kernel import synthetic_scene
from xsarsea_amd import _lib
lco = olut.to_lut("gmf_cmod5n")
co, _ = lut_dicts(lco, None)
ctx = _lib.Context(0)
ctx.upload_luts(co=co)
inc, s_vv, _, _, anc = synthetic_scene(256, 1024, np.float32, 9)
anc = (anc * 2.5).astype(np.complex64)   # a third of the pixels leave the monotone rows: far more than an eighth go to the list
inc[...] = np.where(np.isnan(inc), np.nan, 20.0 + (inc - 30.0) * 0.5)
import torch
dev = torch.device("cuda", 0)
t = [torch.from_numpy(a).to(dev) for a in (inc, s_vv, anc)]
out = torch.empty(inc.shape, dtype=torch.complex64, device=dev)
torch.cuda.synchronize()
ctx.timing_enable(True)
ctx.invert_raw(256, 1024, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, t[0].data_ptr(), t[1].data_ptr(), None, None, t[2].data_ptr(), out.data_ptr(), None, algo=_lib.ALGO_PRUNED)
tm = ctx.timing()
ex = ctx.invert_host(inc, sigma0_co=s_vv, anc=anc, algo="exact", out_dtype=np.complex64)
same = np.array_equal(out.cpu().numpy().view(np.int32), ex[0].view(np.int32))
print("RESULT", tm["launches"], tm["last_list_pixels"], tm["last_band2_pixels"], int(same), tm["last_blocks_pixels"])
"""


@pytest.mark.parametrize("route", ["overflow", "alloc_failure"])
def test_work_list_overflow_and_allocation_failure(route):
    """A scene that leaves more pixels undecided than the capped work lists hold (capacity shrunk to 40 here; an eighth of
    the raster in production): the consumers take their list and then the pixels marked in the strip masks.  A work list that cannot be allocated (XSW_FAIL_LIST_ALLOC=1) selects the one-kernel path
    instead of an error.  Either way the winds equal XSW_ALGO_EXACT's, bit for bit.  (Fresh process: the switches are read once.)"""
    env = dict(os.environ)
    if route == "alloc_failure":
        env["XSW_FAIL_LIST_ALLOC"] = "1"
    else:
        env["XSW_LIST_CAP_TEST"] = "40"  # the three work lists (to k_invert_band2, k_invert_blocks and k_invert_list) hold 40 pixels
    r = subprocess.run([sys.executable, "-c", _OVERFLOW_SCRIPT.format(repo=REPO)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    launches, listed, handed, same, to_blocks = (int(x) for x in line[1:])
    assert same == 1, "winds differ from the exact kernel"
    if route == "overflow":
        # every counter ran past the capacity (what does not fit on list C goes on k_invert_list's list, and on its strip mask from there)
        assert launches == 1 and listed > 40 and handed > 40 and to_blocks > 40, (listed, handed, to_blocks)
    else:
        assert launches == 0  # no two-kernel launch took place


def test_switching_streams_waits_for_the_work_in_flight(gpu_ctx, default_luts):
    """Two inversions on two streams share the context's work list: xsw_set_stream waits for the first before the second may
    reset the list.  Results of both == a run on one stream."""
    torch = pytest.importorskip("torch")
    from xsarsea_amd import _lib
    lco, _ = default_luts
    co, _ = lut_dicts(lco, None)
    gpu_ctx.upload_luts(co=co)
    inc, s_vv, _, _, anc = synthetic_scene(600, 2000, np.float32, 31)
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(a).to(dev) for a in (inc, s_vv, anc)]
    outs = [torch.empty(inc.shape, dtype=torch.complex64, device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    run = lambda o: gpu_ctx.invert_raw(600, 2000, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, t[0].data_ptr(), t[1].data_ptr(), None, None,
                                       t[2].data_ptr(), o.data_ptr(), None, algo=_lib.ALGO_PRUNED)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    try:
        run(outs[0])
        gpu_ctx.synchronize()
        gpu_ctx.set_stream(s1.cuda_stream)
        run(outs[1])
        gpu_ctx.set_stream(s2.cuda_stream)  # must not let the next call's list reset overtake the first call's k_invert_list
        run(outs[2])
        gpu_ctx.synchronize()
        torch.cuda.synchronize()
    finally:
        gpu_ctx.use_own_stream()
    a = torch.view_as_real(outs[0]).view(torch.int32)
    for o in outs[1:]:
        assert torch.equal(a, torch.view_as_real(o).view(torch.int32))


def test_codes_that_are_not_codes_expand_to_nan(gpu_ctx, default_luts):
    """A code whose index lies outside the context's LUT (stale memory, codes of another LUT) expands to (nan, nan) on the
    device and on the host instead of reading outside the tables."""
    torch = pytest.importorskip("torch")
    from xsarsea_amd import _lib
    lco, lcr = default_luts
    co, cr = lut_dicts(lco, lcr)
    gpu_ctx.upload_luts(co=co, cr=cr)
    plane = len(lco.wspd) * len(lco.phi)
    cc = np.array([0, plane - 1, plane, 0x3FFFFFFF, 0x40000000 | (plane - 1), 0x40000000 | plane, 0x80000000, _lib.CODE_NAN, _lib.CODE_NAN_RE], np.uint32)
    ccr = np.array([0, len(lcr.wspd) - 1, len(lcr.wspd), 0x3FFFFFFE, _lib.CODE_NO_INDEX, _lib.CODE_PICK_CO | 5, 0x80000001, _lib.CODE_NAN, _lib.CODE_NAN_RE], np.uint32)
    h_co, h_cr = np.empty(cc.size, np.complex128), np.empty(cc.size, np.complex128)
    gpu_ctx.expand_codes_raw(cc.size, _lib.MEM_HOST, _lib.XSW_F64, cc.ctypes.data, ccr.ctypes.data, h_co.ctypes.data, h_cr.ctypes.data)
    dev = torch.device("cuda", 0)
    t_cc, t_ccr = torch.from_numpy(cc.view(np.int32)).to(dev), torch.from_numpy(ccr.view(np.int32)).to(dev)
    d_co, d_cr = torch.empty(cc.size, dtype=torch.complex128, device=dev), torch.empty(cc.size, dtype=torch.complex128, device=dev)
    gpu_ctx.expand_codes_raw(cc.size, _lib.MEM_DEVICE, _lib.XSW_F64, t_cc.data_ptr(), t_ccr.data_ptr(), d_co.data_ptr(), d_cr.data_ptr())
    gpu_ctx.synchronize()
    assert np.array_equal(_bits(h_co), _bits(d_co.cpu().numpy())) and np.array_equal(_bits(h_cr), _bits(d_cr.cpu().numpy()))
    assert np.array_equal(np.isnan(h_co.real), [False, False, True, True, False, True, True, True, True])
    # (entry 5 picks the co-pol wind, whose own code is out of range: NaN)
    assert np.array_equal(np.isnan(h_cr.real), [False, False, True, True, True, True, True, True, True])
    assert h_cr[0] == 3.0 * np.exp(1j * 0.0) or np.isclose(abs(h_cr[0]), 3.0)
    assert h_co[8].imag == 0.0 and np.isnan(h_co[7].imag)


def test_flat_raster_is_recut_into_tiles(gpu_ctx, default_luts):
    """A 1-D vector of pixels (lines = 1: what stacked / flattened inputs give) is re-cut by the host path into lines of 4096
    samples + a tail, so that the workgroups stay full and the chunks pipeline: same bits as the one-line raster inverted in HBM."""
    torch = pytest.importorskip("torch")
    from xsarsea_amd import _lib
    lco, lcr = default_luts
    co, cr = lut_dicts(lco, lcr)
    gpu_ctx.upload_luts(co=co, cr=cr)
    inc, s_vv, s_vh, dsig, anc = (a.reshape(-1)[:200_003] for a in synthetic_scene(300, 700, np.float32, 29))  # 48 lines of 4096 + 3395
    ref = _device_run(gpu_ctx, torch, _lib, tuple(a.reshape(1, -1) for a in (inc, s_vv, s_vh, dsig, anc)), np.complex128, {"complex", "codes"},
                      dual_select=True)
    # (the device-raster path re-cuts the one-line raster too: compare with 49 separately inverted lines of 4096)
    lines49 = tuple(np.pad(a, (0, 49 * 4096 - a.size), constant_values=np.nan).reshape(49, 4096) for a in (inc, s_vv, s_vh, dsig, anc))
    sep = _device_run(gpu_ctx, torch, _lib, lines49, np.complex128, {"complex"}, dual_select=True)
    assert np.array_equal(_bits(sep["co"].cpu().numpy().reshape(-1)[:inc.size]), _bits(ref["co"].cpu().numpy().reshape(-1)))
    assert np.array_equal(_bits(sep["cr"].cpu().numpy().reshape(-1)[:inc.size]), _bits(ref["cr"].cpu().numpy().reshape(-1)))
    g = gpu_ctx.invert_host(inc, sigma0_co=s_vv, sigma0_cr=s_vh, dsig_cr=dsig, anc=anc, dual_select=True, want_idx=True, want_codes=True,
                            algo="pruned")
    assert g[0].shape == inc.shape
    assert np.array_equal(_bits(g[0]), _bits(ref["co"].cpu().numpy().reshape(-1)))
    assert np.array_equal(_bits(g[1]), _bits(ref["cr"].cpu().numpy().reshape(-1)))
    assert np.array_equal(g[3][0], ref["cc"].cpu().numpy().reshape(-1).view(np.uint32))


def test_staging_callback_contract(gpu_ctx, default_luts):
    """xsw_invert_args.stage: a callback that fills a piece is used instead of the raster pointer (the raster passed for sigma0 is
    never read); returning 0 leaves the default copy in place; an exception inside the callback aborts the call, crosses no C
    frame and is re-raised to the caller -- and the context stays usable."""
    import ctypes
    lco, _ = default_luts
    co, _ = lut_dicts(lco, None)
    gpu_ctx.upload_luts(co=co)
    inc, s_vv, _, _, anc = synthetic_scene(300, 700, np.float32, 37)
    ref = gpu_ctx.invert_host(inc, sigma0_co=s_vv, anc=anc, out_dtype=np.complex64, algo="pruned")
    calls = []
    flat = s_vv.reshape(-1)

    def stage(which, px0, npx, dst):
        calls.append((which, px0, npx))
        if which != 1:
            return 0  # incidence and ancillary wind: the library's own copy
        out = np.frombuffer((ctypes.c_char * (npx * 4)).from_address(dst), dtype=np.float32)
        out[...] = flat[px0:px0 + npx]
        return 1

    junk = np.full_like(s_vv, 123.0)  # must not be read
    got = gpu_ctx.invert_host(inc, sigma0_co=junk, anc=anc, out_dtype=np.complex64, algo="pruned", stage=stage)
    assert np.array_equal(_bits(got[0]), _bits(ref[0]))
    assert {c[0] for c in calls} == {0, 1, 4} and sum(c[2] for c in calls if c[0] == 1) == inc.size

    def boom(which, px0, npx, dst):
        raise ZeroDivisionError("in the staging callback")

    with pytest.raises(ZeroDivisionError, match="staging callback"):
        gpu_ctx.invert_host(inc, sigma0_co=s_vv, anc=anc, out_dtype=np.complex64, stage=boom)
    again = gpu_ctx.invert_host(inc, sigma0_co=s_vv, anc=anc, out_dtype=np.complex64, algo="pruned")
    assert np.array_equal(_bits(again[0]), _bits(ref[0]))
    # page-locked memory of another context / a foreign pointer is refused by xsw_host_free
    from xsarsea_amd import _lib
    with pytest.raises(_lib.XswError, match="not a pointer of xsw_host_alloc"):
        gpu_ctx._check(gpu_ctx._lib.xsw_host_free(gpu_ctx._h, ctypes.c_void_p(got[0].ctypes.data)), "xsw_host_free")


def test_staging_is_released(gpu_ctx, default_luts):
    """the page-locked / device staging of the host workers is not kept for the life of the process -- lowering
    the worker count frees the surplus workers at once, XSW_STAGING_KEEP_MB=0 (fresh process: read once) releases everything after
    each call; results stay those of the default configuration."""
    lco, _ = default_luts
    co, _ = lut_dicts(lco, None)
    gpu_ctx.upload_luts(co=co)
    inc, s_vv, _, _, anc = synthetic_scene(1024, 4096, np.float32, 3)  # 4.2 Mpx: sixteen chunks, every worker gets one
    ref = gpu_ctx.invert_host(inc, sigma0_co=s_vv, anc=anc, out_dtype=np.complex64)[0]
    try:
        gpu_ctx.set_host_threads(2)   # ten workers' buffers go back now
        a = gpu_ctx.invert_host(inc, sigma0_co=s_vv, anc=anc, out_dtype=np.complex64)[0]
        gpu_ctx.set_host_threads(12)  # ... and are pinned again on demand
        b = gpu_ctx.invert_host(inc, sigma0_co=s_vv, anc=anc, out_dtype=np.complex64)[0]
    finally:
        gpu_ctx.set_host_threads(0)
    assert np.array_equal(a.view(np.int32), ref.view(np.int32)) and np.array_equal(b.view(np.int32), ref.view(np.int32))
    script = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {REPO!r}); sys.path.insert(0, {REPO!r} + '/tests')\n"
        "from oracle import lut as olut\nfrom util import lut_dicts\nfrom test_gpu_kernel import synthetic_scene\nfrom xsarsea_amd import _lib\n"
        "ctx = _lib.Context(0); ctx.upload_luts(co=lut_dicts(olut.to_lut('gmf_cmod5n', resolution='low'), None)[0])\n"
        "inc, s, _, _, anc = synthetic_scene(512, 4096, np.float32, 3)\n"
        "r = [ctx.invert_host(inc, sigma0_co=s, anc=anc, out_dtype=np.complex64)[0] for _ in range(3)]\n"
        "print('SAME', int(all(np.array_equal(r[0].view(np.int32), x.view(np.int32)) for x in r[1:])))\n")
    r = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, XSW_STAGING_KEEP_MB="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SAME 1" in r.stdout, r.stderr[-1500:]
