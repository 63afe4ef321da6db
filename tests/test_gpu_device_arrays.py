"""GPU: the drop-in API on rasters that already live in HBM (torch CUDA tensors / __cuda_array_interface__ objects) and
the single-process multi-device path (`options.devices`), both against the oracle through the numpy route they must equal."""
import warnings

import numpy as np
import pytest

from test_gpu_kernel import synthetic_scene
from util import assert_complex_close, bits_equal

pytestmark = pytest.mark.gpu


def _oracle(inc, s_vv, s_vh, dsig, anc, luts, mode):
    from oracle import invert as oinv
    lco, lcr = luts
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if mode == "mono":
            return oinv.invert_from_model(inc, s_vv, ancillary_wind=anc, lut_co=lco)
        if mode == "dual":
            return oinv.invert_from_model(inc, s_vv, s_vh, ancillary_wind=anc, dsig_cr=dsig, lut_co=lco, lut_cr=lcr)
        return oinv.invert_from_model(inc, s_vh, dsig_cr=0.1, lut_cr=lcr)


class _CaiOnly:
    """An object that is NOT a torch tensor and only exposes __cuda_array_interface__ (what cupy / numba arrays look like)."""

    def __init__(self, t):
        self._t = t
        self.__cuda_array_interface__ = t.__cuda_array_interface__


def test_invert_from_model_device_tensors_float64(gpu_ctx, lowres_luts):
    """float64 torch tensors in -> complex128 torch tensors out on the same device; mono, dual (fused select) and mono
    cross-pol equal the oracle (same dB arithmetic on both sides for float64: indices identical, values to 1e-12)."""
    torch = pytest.importorskip("torch")
    from xsarsea_amd import windspeed
    from test_gpu_kernel import assert_dual_select
    dev = torch.device("cuda", 0)
    inc, s_vv, s_vh, dsig, anc = synthetic_scene(64, 200, np.float64, 41)
    t = [torch.from_numpy(a).to(dev) for a in (inc, s_vv, s_vh, dsig, anc)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        co = windspeed.invert_from_model(t[0], t[1], ancillary_wind=t[4], model="gmf_cmod5n", resolution="low")
        co2, dual = windspeed.invert_from_model(t[0], t[1], t[2], ancillary_wind=t[4], dsig_cr=t[3], model=("gmf_cmod5n", "gmf_s1_v2"),
                                                resolution="low")
        cr = windspeed.invert_from_model(_CaiOnly(t[0]), _CaiOnly(t[2]), dsig_cr=0.1, model="gmf_s1_v2", resolution="low")
    for x in (co, co2, dual):
        assert isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.complex128 and tuple(x.shape) == inc.shape
    assert isinstance(cr, torch.Tensor) and cr.dtype == torch.float64
    torch.cuda.synchronize()
    o_co = _oracle(inc, s_vv, s_vh, dsig, anc, lowres_luts, "mono")
    o_co2, o_dual = _oracle(inc, s_vv, s_vh, dsig, anc, lowres_luts, "dual")
    o_cr = _oracle(inc, s_vv, s_vh, dsig, anc, lowres_luts, "cross")
    assert_complex_close(co.cpu().numpy(), o_co, what="device mono")
    assert_complex_close(co2.cpu().numpy(), o_co2, what="device dual co")
    from oracle import invert as oinv
    raw = oinv.invert_numpy(oinv.Prepared(*lowres_luts), inc, oinv.to_db(s_vv), oinv.to_db(s_vh), dsig, anc)[1]
    assert_dual_select(dual.cpu().numpy(), o_dual, raw, "device dual select")
    assert np.allclose(cr.cpu().numpy(), o_cr, rtol=1e-12, atol=0, equal_nan=True)
    # the numpy route on the same rasters is bit-identical for float64 (both convert to dB on the device)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        co_np = windspeed.invert_from_model(inc, s_vv, ancillary_wind=anc, model="gmf_cmod5n", resolution="low")
    assert bits_equal(co.cpu().numpy(), co_np)


def test_invert_from_model_device_tensors_float32_and_mixed(gpu_ctx, lowres_luts):
    """float32 device rasters (the benchmark's dtype): complex64 on request, 1-D incidence row broadcast, a host numpy
    ancillary wind mixed in (uploaded); equals the numpy route with the device dB conversion, bit for bit."""
    torch = pytest.importorskip("torch")
    import xsarsea_amd
    from xsarsea_amd import windspeed
    dev = torch.device("cuda", 0)
    inc, s_vv, _, _, anc = synthetic_scene(50, 300, np.float32, 43)
    inc_row = inc[7].copy()
    t_inc, t_s = torch.from_numpy(inc_row).to(dev), torch.from_numpy(s_vv).to(dev)
    old = (xsarsea_amd.options.device_out_dtype, xsarsea_amd.options.db_on_device)
    try:
        xsarsea_amd.options.device_out_dtype = "complex64"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            co = windspeed.invert_from_model(t_inc, t_s, ancillary_wind=anc, model="gmf_cmod5n", resolution="low")
            xsarsea_amd.options.db_on_device = True
            ref = windspeed.invert_from_model(inc_row, s_vv, ancillary_wind=anc, model="gmf_cmod5n", resolution="low")
    finally:
        xsarsea_amd.options.device_out_dtype, xsarsea_amd.options.db_on_device = old
    assert co.dtype == torch.complex64 and tuple(co.shape) == s_vv.shape
    assert bits_equal(co.cpu().numpy(), ref.astype(np.complex64))
    # assertion of the reference: a co-pol inversion needs a valid ancillary wind -- also for device rasters
    with pytest.raises(AssertionError):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            windspeed.invert_from_model(t_inc, t_s, ancillary_wind=torch.full(s_vv.shape, float("nan"), dtype=torch.complex64, device=dev),
                                        model="gmf_cmod5n", resolution="low")


def test_detrend_and_nesz_on_device_tensors(gpu_ctx):
    torch = pytest.importorskip("torch")
    import xsarsea_amd
    from xsarsea_amd.windspeed import nesz_flattening
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    inc = np.broadcast_to(np.linspace(20, 45, 300), (120, 300)).copy()
    sig = rng.gamma(2.0, 0.01, inc.shape)
    det_np = xsarsea_amd.sigma0_detrend(sig, inc)
    det = xsarsea_amd.sigma0_detrend(torch.from_numpy(sig).to(dev), torch.from_numpy(inc).to(dev))
    assert isinstance(det, torch.Tensor) and det.is_cuda and det.dtype == torch.float64
    assert bits_equal(det.cpu().numpy(), det_np)
    old = xsarsea_amd.options.nesz_on_device
    try:
        xsarsea_amd.options.nesz_on_device = "device"
        nz_np = nesz_flattening(sig, inc)
    finally:
        xsarsea_amd.options.nesz_on_device = old
    nz = nesz_flattening(torch.from_numpy(sig).to(dev), torch.from_numpy(inc).to(dev))
    assert isinstance(nz, torch.Tensor) and bits_equal(nz.cpu().numpy(), nz_np)
    # float32 rasters that start 4 bytes off a 16-byte boundary (contiguous views into a larger buffer): the kernels' element-wise
    # loads must give the bits of the aligned vector loads
    s32, i32 = torch.from_numpy(sig.astype(np.float32)).to(dev), torch.from_numpy(inc.astype(np.float32)).to(dev)
    n = s32.numel()
    off_s, off_i = torch.empty(n + 1, dtype=torch.float32, device=dev)[1:].view(s32.shape), torch.empty(n + 1, dtype=torch.float32, device=dev)[1:].view(s32.shape)
    off_s.copy_(s32)
    off_i.copy_(i32)
    assert off_s.data_ptr() % 16 == 4 and off_s.is_contiguous()
    assert bits_equal(nesz_flattening(off_s, off_i).cpu().numpy(), nesz_flattening(s32, i32).cpu().numpy())
    assert bits_equal(xsarsea_amd.sigma0_detrend(off_s, off_i).cpu().numpy(), xsarsea_amd.sigma0_detrend(s32, i32).cpu().numpy())


@pytest.mark.parametrize("mode", ["mono", "dual"])
def test_single_process_multi_device(gpu_ctx, lowres_luts, mode):
    """`options.devices`: row tiles of one host raster inverted by several contexts from host threads, results landed in
    place.  On the one-GPU box the list names device 0 three times (three contexts, three threads, uneven tiles): the result
    must equal the one-context call bit for bit -- and so the oracle."""
    import xsarsea_amd
    from xsarsea_amd import windspeed
    inc, s_vv, s_vh, dsig, anc = synthetic_scene(203, 130, np.float32, 47)
    anc[:70] = np.nan  # a whole tile without ancillary wind: NaN rows, no per-tile assertion
    model = "gmf_cmod5n" if mode == "mono" else ("gmf_cmod5n", "gmf_s1_v2")
    args = (inc, s_vv) if mode == "mono" else (inc, s_vv, s_vh)
    kw = dict(ancillary_wind=anc, model=model, resolution="low")
    if mode == "dual":
        kw["dsig_cr"] = dsig
    old = (xsarsea_amd.options.devices, xsarsea_amd.options.devices_min_pixels)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            one = windspeed.invert_from_model(*args, **kw)
            xsarsea_amd.options.devices, xsarsea_amd.options.devices_min_pixels = [0, 0, 0], 0
            many = windspeed.invert_from_model(*args, **kw)
    finally:
        xsarsea_amd.options.devices, xsarsea_amd.options.devices_min_pixels = old
    one, many = (x if isinstance(x, tuple) else (x,) for x in (one, many))
    for a, b in zip(one, many):
        assert bits_equal(a, b)
    assert np.isnan(many[0][:70]).all() and not np.isnan(many[0][70:, 3:]).all()


def test_device_rasters_with_sigma0_from_the_host(gpu_ctx, default_luts):
    """XSW_MEM_DEVICE_SIGMA0_HOST (round 4): incidence / ancillary wind / outputs resident in HBM, the linear float32 sigma0 rasters
    on the host; the staging callback converts each chunk to dB with numpy's own log10 inside the library's worker ring.  Winds ==
    the all-device call on the host-converted dB rasters, bit for bit -- mono and dual-pol (fused select), ragged last chunk."""
    import ctypes
    torch = pytest.importorskip("torch")
    from xsarsea_amd import _lib
    from test_gpu_kernel import synthetic_scene
    from util import lut_dicts
    lco, lcr = default_luts
    co, cr = lut_dicts(lco, lcr)
    gpu_ctx.upload_luts(co=co, cr=cr)
    lines, samples = 2050, 1500  # ~3.1 Mpx: two chunks of the worker ring, the second one ragged
    inc, s_vv, s_vh, dsig, anc = synthetic_scene(lines, samples, np.float32, 5)
    dev = torch.device("cuda", 0)
    t_inc, t_dsig, t_anc = (torch.from_numpy(a).to(dev) for a in (inc, dsig, anc))
    with np.errstate(all="ignore"):
        db_vv, db_vh = (10 * np.log10(s_vv + 1e-15)).astype(np.float32), (10 * np.log10(s_vh + 1e-15)).astype(np.float32)
    t_vv, t_vh = torch.from_numpy(db_vv).to(dev), torch.from_numpy(db_vh).to(dev)
    src = {_lib.STAGE_SIGMA0_CO: s_vv.reshape(-1), _lib.STAGE_SIGMA0_CR: s_vh.reshape(-1)}
    calls = []

    def stage(which, px0, npx, dst):
        if which not in src:
            return 0
        calls.append((which, px0, npx))
        o = np.frombuffer((ctypes.c_char * (npx * 4)).from_address(dst), dtype=np.float32)
        with np.errstate(all="ignore"):
            o[...] = 10 * np.log10(src[which][px0:px0 + npx] + 1e-15)
        return 1

    for dual in (False, True):
        ref_co = torch.empty((lines, samples), dtype=torch.complex64, device=dev)
        ref_cr = torch.empty_like(ref_co) if dual else None
        got_co, got_cr = torch.zeros_like(ref_co), (torch.zeros_like(ref_co) if dual else None)
        p = lambda t: None if t is None else t.data_ptr()
        gpu_ctx.invert_raw(lines, samples, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE, p(t_inc), p(t_vv), p(t_vh) if dual else None,
                           p(t_dsig) if dual else None, p(t_anc), p(ref_co), p(ref_cr), sigma0_is_db=True, dual_select=dual)
        gpu_ctx.synchronize()
        calls.clear()
        gpu_ctx.invert_raw(lines, samples, _lib.XSW_F32, _lib.XSW_F32, _lib.MEM_DEVICE_SIGMA0_HOST, p(t_inc), s_vv.ctypes.data,
                           s_vh.ctypes.data if dual else None, p(t_dsig) if dual else None, p(t_anc), p(got_co), p(got_cr),
                           sigma0_is_db=True, dual_select=dual, stage=stage)
        bits = lambda t: torch.view_as_real(t).view(torch.int32)
        assert torch.equal(bits(got_co), bits(ref_co))
        if dual:
            assert torch.equal(bits(got_cr), bits(ref_cr))
        staged = sorted(c for c in calls if c[0] == _lib.STAGE_SIGMA0_CO)
        assert len(staged) >= 2 and sum(c[2] for c in staged) == lines * samples and staged[0][1] == 0
