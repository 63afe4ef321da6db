"""GPU: the drop-in Python API (numpy in -> numpy out) against the oracle's restatement of the
reference's `invert_from_model` / `sigma0_detrend`; reads like the reference's own test_inversion."""
import os
import warnings

import numpy as np
import pytest

from util import assert_complex_close, bits_equal
from test_gpu_kernel import synthetic_scene

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def scene():
    return synthetic_scene(40, 90, np.float64, 31)


def _oracle(inc, s_vv, s_vh, dsig, anc, lowres_luts, mode):
    from oracle import invert as oinv
    lco, lcr = lowres_luts
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if mode == "mono":
            return oinv.invert_from_model(inc, s_vv, ancillary_wind=anc, lut_co=lco)
        if mode == "dual":
            return oinv.invert_from_model(inc, s_vv, s_vh, ancillary_wind=anc, dsig_cr=dsig, lut_co=lco, lut_cr=lcr)
        return oinv.invert_from_model(inc, s_vh, dsig_cr=0.1, lut_cr=lcr)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_invert_from_model_numpy(gpu_ctx, lowres_luts, scene, dtype):
    """mono co-pol, dual-pol, mono cross-pol: same call shapes as test_xsarsea.py:113-122; results must be
    BIT-identical to the CPU path (complex128 / float64), float32 inputs included (host dB, options 'auto')."""
    from xsarsea_amd import windspeed
    cdt = np.complex64 if dtype == np.float32 else np.complex128
    inc, s_vv, s_vh, dsig, anc = (a.astype(t) for a, t in zip(scene, (dtype,) * 4 + (cdt,)))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        co = windspeed.invert_from_model(inc, s_vv, ancillary_wind=anc, model="gmf_cmod5n", resolution="low")
        co2, dual = windspeed.invert_from_model(inc, s_vv, s_vh, ancillary_wind=anc, dsig_cr=dsig,
                                                model=("gmf_cmod5n", "gmf_s1_v2"), resolution="low")
        cr = windspeed.invert_from_model(inc, s_vh, dsig_cr=0.1, model="gmf_s1_v2", resolution="low")
    assert isinstance(co, np.ndarray) and co.dtype == np.complex128 and co.shape == inc.shape
    assert isinstance(dual, np.ndarray) and cr.dtype == np.float64
    o_co = _oracle(inc, s_vv, s_vh, dsig, anc, lowres_luts, "mono")
    o_co2, o_dual = _oracle(inc, s_vv, s_vh, dsig, anc, lowres_luts, "dual")
    o_cr = _oracle(inc, s_vv, s_vh, dsig, anc, lowres_luts, "cross")
    assert bits_equal(co, o_co) and bits_equal(co2, o_co2)
    assert bits_equal(dual, o_dual)
    assert bits_equal(cr, o_cr)


def test_scalar_dsig_and_kwargs(gpu_ctx, scene):
    """`dsig_cr` scalar broadcast (windspeed.py:122-123) and LUT kwargs forwarding (`inc_step_lr`)."""
    from oracle import invert as oinv, lut as olut
    from xsarsea_amd import windspeed
    inc, s_vv, s_vh, dsig, anc = (a[:10] for a in scene)
    kw = dict(resolution="low", inc_step_lr=2.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        co, dual = windspeed.invert_from_model(inc, s_vv, s_vh, ancillary_wind=anc, dsig_cr=0.3, dsig_co=0.2,
                                               model=("cmod5n", "s1_v2"), **kw)
    lco, lcr = olut.to_lut("gmf_cmod5n", **kw), olut.to_lut("gmf_s1_v2", **kw)
    assert lco.values.shape[0] == 26
    o_co, o_dual = oinv.invert_from_model(inc, s_vv, s_vh, ancillary_wind=anc, dsig_cr=0.3, dsig_co=0.2, lut_co=lco, lut_cr=lcr)
    assert bits_equal(co, o_co) and bits_equal(dual, o_dual)


def test_sigma0_detrend(gpu_ctx):
    """Config 1 of BASELINE.json: 1024 x 1024 VV sigma0 + incidence, CMOD5.N."""
    import xsarsea_amd
    from oracle import detrend as odet
    rng = np.random.default_rng(11)
    inc = np.broadcast_to(np.linspace(30, 46, 1024), (1024, 1024)) + 0.02 * rng.standard_normal((1024, 1))
    for dt in (np.float64, np.float32):
        s = rng.uniform(0.005, 0.3, inc.shape).astype(dt)
        out = xsarsea_amd.sigma0_detrend(s, inc.astype(dt))
        ref = odet.sigma0_detrend(s, inc.astype(dt))
        assert out.dtype == np.float64 and out.shape == s.shape
        assert np.allclose(out, ref, rtol=1e-13, atol=0)
    with pytest.raises(ValueError):
        xsarsea_amd.sigma0_detrend(s, inc, wind_speed_gmf=np.array([5.0, 10.0]))


def _kw():
    return dict(model=("gmf_cmod5n", "gmf_s1_v2"), resolution="low")


def test_xarray_containers(gpu_ctx, scene, xr_env):
    """xarray in -> xarray out (windspeed.py:337-343, :395-438 of the reference): name `windspeed_gmf`, dims/coords of the
    inputs, `comment` / `model` (/ `units`) attrs per routing, values bit-identical to the numpy path."""
    xr = xr_env.xr
    from xsarsea_amd import windspeed
    arrs = [a[:8] for a in scene]
    coords = {"line": np.arange(8) * 10, "sample": np.arange(arrs[0].shape[1]) * 10}
    inc, s_vv, s_vh, dsig, anc = (xr.DataArray(a, dims=("line", "sample"), coords=coords, attrs={"stale": 1}) for a in arrs)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        n_co, n_dual = windspeed.invert_from_model(*arrs[:3], ancillary_wind=arrs[4], dsig_cr=arrs[3], **_kw())
        n_cr = windspeed.invert_from_model(arrs[0], arrs[2], dsig_cr=0.1, model="gmf_s1_v2", resolution="low")
        co, dual = windspeed.invert_from_model(inc, s_vv, s_vh, ancillary_wind=anc, dsig_cr=dsig, **_kw())
        mono = windspeed.invert_from_model(inc, s_vv, ancillary_wind=anc, model="gmf_cmod5n", resolution="low")
        cr = windspeed.invert_from_model(inc, s_vh, dsig_cr=0.1, model="gmf_s1_v2", resolution="low")
    for out in (co, dual, mono, cr):
        assert isinstance(out, xr.DataArray) and tuple(out.dims) == ("line", "sample") and out.shape == arrs[0].shape
        assert "stale" not in out.attrs  # attrs.clear() (:340)
    assert co.name == "windspeed_gmf" and mono.name == "windspeed_gmf"
    assert co.attrs == {"comment": "wind speed and direction inverted from model gmf_cmod5n (VV)", "model": "gmf_cmod5n"}
    assert mono.attrs == co.attrs
    assert dual.attrs == {"comment": "wind speed and direction inverted from model gmf_cmod5n (VV) and gmf_s1_v2 (VH)",
                          "model": "gmf_cmod5n gmf_s1_v2"}
    assert cr.attrs == {"comment": "wind speed inverted from model gmf_s1_v2 (VH)", "model": "gmf_s1_v2", "units": "m/s"}
    assert co.dtype == np.complex128 and dual.dtype == np.complex128 and cr.dtype == np.float64
    assert bits_equal(np.asarray(co.values), n_co) and bits_equal(np.asarray(dual.values), n_dual)
    assert bits_equal(np.asarray(mono.values), n_co) and bits_equal(np.asarray(cr.values), n_cr)
    assert np.array_equal(np.asarray(co["line"] if not xr_env.standin else co.coords["line"]), coords["line"])


def test_xarray_pol_check(gpu_ctx, scene, xr_env):
    """`sigma0.pol` is checked against the model's polarisation for mono-pol calls (windspeed.py:88-101): mismatch ->
    ValueError, match -> no "Unable to check" warning, absent -> UserWarning."""
    xr = xr_env.xr
    from xsarsea_amd import windspeed
    inc, s_vv, _, _, anc = (a[:4] for a in scene)
    mk = lambda a, **c: xr.DataArray(a, dims=("line", "sample"), coords=c)
    with pytest.raises(ValueError, match="sigma0 pol is VH, and model gmf_cmod5n can only handle VV"):
        windspeed.invert_from_model(mk(inc), mk(s_vv, pol="VH"), ancillary_wind=mk(anc), model="gmf_cmod5n", resolution="low")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        out = windspeed.invert_from_model(mk(inc), mk(s_vv, pol="VV"), ancillary_wind=mk(anc), model="gmf_cmod5n", resolution="low")
    assert isinstance(out, xr.DataArray)
    with pytest.warns(UserWarning, match="Unable to check sigma0 pol"):
        windspeed.invert_from_model(mk(inc), mk(s_vv), ancillary_wind=mk(anc), model="gmf_cmod5n", resolution="low")
    with pytest.raises(AssertionError):  # co-pol inversion without any valid ancillary wind (:107)
        windspeed.invert_from_model(mk(inc), mk(s_vv, pol="VV"), ancillary_wind=mk(anc * np.nan), model="gmf_cmod5n", resolution="low")


def test_dask_blocks_are_lazy_and_equal_the_numpy_path(gpu_ctx, scene, xr_env):
    """dask-backed DataArrays (row chunks, sample axis whole: windspeed.py:350-364): the call returns lazily -- no device
    work before .compute() -- runs once per row block, and gives the numpy path's bits."""
    xr, da = xr_env.xr, xr_env.da
    from xsarsea_amd import _lib, windspeed
    arrs = [a[:24] for a in scene]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        n_co, n_dual = windspeed.invert_from_model(*arrs[:3], ancillary_wind=arrs[4], dsig_cr=arrs[3], **_kw())
    calls = []
    real = _lib.Context.invert_host

    def counting(self, *a, **k):
        calls.append(np.shape(a[0]))
        return real(self, *a, **k)

    lazy = [xr.DataArray(da.from_array(a, chunks=(7, -1)), dims=("line", "sample")) for a in arrs]
    try:
        _lib.Context.invert_host = counting
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            co, dual = windspeed.invert_from_model(lazy[0], lazy[1], lazy[2], ancillary_wind=lazy[4], dsig_cr=lazy[3], **_kw())
        assert isinstance(co, xr.DataArray) and isinstance(co.data, da.Array) and isinstance(dual.data, da.Array)
        assert co.name == "windspeed_gmf" and co.attrs["model"] == "gmf_cmod5n" and dual.attrs["model"] == "gmf_cmod5n gmf_s1_v2"
        assert calls == [], "the dask path must not touch the device before compute()"
        v_co, v_dual = np.asarray(co.data.compute()), np.asarray(dual.data.compute())
    finally:
        _lib.Context.invert_host = real
    assert sorted(c[0] for c in calls) == [3, 7, 7, 7] and all(c[1] == arrs[0].shape[1] for c in calls)
    assert bits_equal(v_co, n_co) and bits_equal(v_dual, n_dual)
    # raw dask arrays (no xarray container): the reference falls through to its numpy path and returns numpy
    raw = [da.from_array(a, chunks=(7, -1)) for a in arrs]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r_co, r_dual = windspeed.invert_from_model(raw[0], raw[1], raw[2], ancillary_wind=raw[4], dsig_cr=raw[3], **_kw())
    assert isinstance(r_co, np.ndarray) and isinstance(r_dual, np.ndarray)
    assert bits_equal(r_co, n_co) and bits_equal(r_dual, n_dual)


def test_loop_dimensions_broadcast(gpu_ctx, scene):
    """The gufunc broadcasts its loop dimensions over all inputs (windspeed.py:307-322): a 1-D incidence row against 2-D
    sigma0 / ancillary rasters gives (line, sample) outputs."""
    from xsarsea_amd import windspeed
    inc, s_vv, _, _, anc = (a[:6] for a in scene)
    row = inc[0].copy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = windspeed.invert_from_model(row, s_vv, ancillary_wind=anc, model="gmf_cmod5n", resolution="low")
        b = windspeed.invert_from_model(np.broadcast_to(row, s_vv.shape).copy(), s_vv, ancillary_wind=anc, model="gmf_cmod5n",
                                        resolution="low")
    assert a.shape == s_vv.shape and bits_equal(a, b)


def test_sigma0_detrend_xarray(gpu_ctx, scene, xr_env):
    """DataArray in -> DataArray out with the `comment` attr (detrend.py:55-66); incidence row taken with isel(line=0)."""
    xr = xr_env.xr
    import xsarsea_amd
    inc, s_vv = scene[0][:16], scene[1][:16]
    inc = inc.copy()
    inc[:, :3] = 30.0  # the scene's leading NaN incidence columns: keep the GMF row finite for an exact comparison
    ref = xsarsea_amd.sigma0_detrend(s_vv, inc)
    out = xsarsea_amd.sigma0_detrend(xr.DataArray(s_vv, dims=("line", "sample"), attrs={"units": "linear"}),
                                     xr.DataArray(inc, dims=("line", "sample")))
    assert isinstance(out, xr.DataArray) and tuple(out.dims) == ("line", "sample") and out.dtype == np.float64
    assert out.attrs["comment"] == "detrended with model gmf_cmod5n"
    assert np.array_equal(np.asarray(out.values), ref, equal_nan=True)


def test_nesz_flatten_device(gpu_ctx):
    """xsw_nesz_flatten (windspeed/utils.py:94-163 on the device) against the REFERENCE's outputs (crosspol_prep.npz) and,
    at a larger ragged size, against the oracle.  Tolerances: float64 rasters 1e-10 relative (numpy's polyfit solves by SVD,
    the kernel by the normal equations about the mean abscissa, both in float64: observed ~1e-13); float32 rasters 1e-5
    (numpy takes log10 and the column sums in float32, the kernel in float64)."""
    import warnings
    from conftest import golden
    from oracle import crosspol as ocp
    from xsarsea_amd import options, windspeed
    d = golden("crosspol_prep.npz")
    rel = lambda a, b: float(np.nanmax(np.abs(a - b) / np.abs(b)))
    got = gpu_ctx.nesz_flatten_host(d["nesz_noise"], d["nesz_inc"])
    assert got.dtype == np.float64 and np.isfinite(got).all() and rel(got, d["nesz_flat"]) <= 1e-10
    got32 = gpu_ctx.nesz_flatten_host(d["nesz_noise"].astype(np.float32), d["nesz_inc"].astype(np.float32))
    assert rel(got32, d["nesz_flat32"]) <= 1e-5
    assert np.isnan(gpu_ctx.nesz_flatten_host(np.full((3, 8), np.nan), d["nesz_inc"][:3, :8])).all()
    # larger, ragged (samples not a multiple of 4 or 256; more line blocks than one), NaN columns/lines, through the API
    rng = np.random.default_rng(8)
    L, S = 1537, 4101
    inc = np.linspace(29.0, 46.0, S)[None, :] + 0.02 * np.sin(np.arange(L) / 90.0)[:, None]
    noise = 10 ** ((-31.0 + 0.1 * (inc - 29.0) + 0.4 * rng.standard_normal((L, S))) / 10.0)
    noise[rng.random((L, S)) < 0.02] = np.nan
    noise[:, 77] = np.nan
    noise[500, :] = np.nan
    noise[3, 9] = 0.0
    old = (options.nesz_on_device, options.nesz_device_min_size)
    try:
        options.nesz_on_device = "device"
        dev = windspeed.nesz_flattening(noise, inc)
        options.nesz_on_device, options.nesz_device_min_size = "auto", 1 << 20
        assert np.array_equal(windspeed.nesz_flattening(noise, inc), dev)  # 6.3e6 px: "auto" takes the device too
    finally:
        options.nesz_on_device, options.nesz_device_min_size = old
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = ocp.nesz_flattening(noise, inc)
    assert dev.shape == ref.shape and np.isfinite(dev).all() and rel(dev, ref) <= 1e-10, rel(dev, ref)
    # the same ragged raster in float32 (odd line length: the element-wise paths of the fit and the write pass), and an even
    # crop of it that starts 4 bytes off a 16-byte boundary on the host (the device copies are aligned: the vector paths)
    n32, i32 = noise.astype(np.float32), inc.astype(np.float32)
    for a, b in ((n32, i32), (np.ascontiguousarray(n32[1:, 1:]), np.ascontiguousarray(i32[1:, 1:]))):
        got = gpu_ctx.nesz_flatten_host(a, b)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref32 = ocp.nesz_flattening(a, b)
        assert got.dtype == np.float64 and got.shape == ref32.shape and np.isfinite(got).all() and rel(got, ref32) <= 1e-5, rel(got, ref32)
    # one valid column only: every line fits a single point -> numpy's minimum-norm solution (slope = y/2x, icpt = y/2)
    one = np.full((4, 6), np.nan)
    one[:, 2] = [1e-3, 2e-3, 5e-4, 1e-3]
    inc6 = np.broadcast_to(np.linspace(30.0, 35.0, 6), (4, 6)).copy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref1 = ocp.nesz_flattening(one, inc6)
    assert rel(gpu_ctx.nesz_flatten_host(one, inc6), ref1) <= 1e-10


def test_device_built_lut_vs_oracle(gpu_ctx, default_luts, lowres_luts):
    """xsw_lut_build (GMF grid fill + low->high interpolation + dB + search layout, all on the device) against the ORACLE's
    LUT (oracle.lut.to_lut), entry by entry: <= 1e-10 dB (device libm vs numpy; observed ~1e-12), and at index level: the
    reference-generated default goldens inverted on the device-built tables give the reference's winds."""
    from conftest import golden
    from xsarsea_amd import _lib, options, windspeed
    from xsarsea_amd.windspeed import _engine, get_model
    ctx = _lib.Context(0)
    try:
        for res, (lco, lcr) in (("high", default_luts), ("low", lowres_luts)):
            kw = {} if res == "high" else {"resolution": "low"}
            for name, ref, cross in (("gmf_cmod5n", lco, False), ("gmf_s1_v2", lcr, True)):
                plan = get_model(name).device_lut_plan(**kw)
                assert plan is not None
                dl = _engine.DeviceLut(name, plan[0], plan[1], plan[2], key=None)
                assert dl.shape == ref.values.shape
                assert all(np.array_equal(a, b) for a, b in zip((dl.incidence, dl.wspd), (ref.incidence, ref.wspd)))
                dl.build(ctx)
                dev = ctx.read_lut(dl.shape, cross=cross)
                assert np.isfinite(dev).all() and float(np.max(np.abs(dev - ref.values))) <= 1e-10, (res, name)
        # a model without a device plan keeps the host route
        assert get_model("gmf_dummy").device_lut_plan() is None if "gmf_dummy" in windspeed.available_models().index else True
    finally:
        ctx.close()
    d = golden("kernel_default_f64.npz")
    old = options.lut_build
    try:
        options.lut_build = "device"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mono = windspeed.invert_from_model(d["inc"], d["sigma0_vv"], ancillary_wind=d["anc"], model="gmf_cmod5n")
            co, dual = windspeed.invert_from_model(d["inc"], d["sigma0_vv"], d["sigma0_vh"], ancillary_wind=d["anc"],
                                                   dsig_cr=d["dsig_cr"], model=("gmf_cmod5n", "gmf_s1_v2"))
            cr = windspeed.invert_from_model(d["inc"], d["sigma0_vh"], dsig_cr=0.1, model="gmf_s1_v2")
        ctx0 = _lib.default_context(options.device)
        assert isinstance(ctx0.lut_key[0], _engine.DeviceLut) and isinstance(ctx0.lut_key[1], _engine.DeviceLut)
    finally:
        options.lut_build = old
    assert_complex_close(mono, d["mono_co"], what="device-built LUT, mono")
    assert_complex_close(co, d["dual_co"], what="device-built LUT, dual co")
    assert_complex_close(dual, d["dual_dual"], what="device-built LUT, dual")
    assert_complex_close(cr, d["cross_only"], what="device-built LUT, cross only")


def test_lut_interp_device_equals_host(gpu_ctx):
    """the low->high LUT interpolation on the device is bit-identical to the host numpy path
    (three sequential interp1d passes), co-pol 3-D and cross-pol 2-D, and rejects out-of-range targets."""
    import time
    from xsarsea_amd import _lib, options, windspeed
    from xsarsea_amd.windspeed.lut import axis_grid, lerp_axis
    for name in ("gmf_cmod5n", "gmf_s1_v2"):
        m = windspeed.get_model(name)
        raw = m._raw_lut()
        inc, wspd = axis_grid(m.inc_range, 0.1), axis_grid(m.wspd_range, 0.1)
        phi = axis_grid(m.phi_range, 1.0)
        t0 = time.perf_counter()
        host = lerp_axis(lerp_axis(raw.values, raw.incidence, inc, 0), raw.wspd, wspd, 1)
        if phi is not None:
            host = lerp_axis(host, raw.phi, phi, 2)
        t1 = time.perf_counter()
        dev = gpu_ctx.lut_interp(raw.values, raw.incidence, raw.wspd, raw.phi, inc, wspd, phi)
        t2 = time.perf_counter()
        assert dev.shape == host.shape and np.array_equal(dev, host), name
        # and against the ORACLE's restatement of xarray's interp (scipy interp1d per dimension: oracle/lut.py), not
        # only the product's own host path: same raw table in, same bits out
        from oracle import lut as olut
        o_raw = olut.raw_lut(name)
        assert np.array_equal(o_raw.values, raw.values), name
        o_hi = olut.normalize_lut(o_raw, resolution="high")
        assert np.array_equal(dev, o_hi.values), name
        print(f"{name}: host lerp {t1 - t0:.2f} s, device {t2 - t1:.3f} s")
    with pytest.raises(_lib.XswError, match="outside the interpolation range"):
        gpu_ctx.lut_interp(raw.values, raw.incidence, raw.wspd, None, inc + 1.0, wspd, None)
    # and the model layer really takes the device route when a GPU is present
    assert options.lut_interp == "auto" and _lib.device_count_safe() > 0


def test_forward_gmf_device(gpu_ctx):
    """the 13 built-in GMFs evaluated on the device agree with the oracle's restatement of the
    reference formulas (device libm vs numpy: ~1e-14 relative, no bitwise claim), incl. the CMOD5 branches."""
    from oracle import gmf as ogmf
    from xsarsea_amd import _lib, options, windspeed
    rng = np.random.default_rng(17)
    n = 200_000
    inc = rng.uniform(16, 66, n)
    for name, (f, pol, wr, pr) in ogmf.GMFS.items():
        wspd = rng.uniform(wr[0], wr[1], n)
        wspd[:1000] = np.linspace(wr[0], wr[1], 1000)   # dense sweep through the low-wind branches
        phi = rng.uniform(-360, 360, n) if pr is not None else None
        with np.errstate(all="ignore"):
            ref = f(inc, wspd, phi)
        got = gpu_ctx.gmf_eval(_lib.GMF_IDS[name], inc, wspd, phi)
        ok = np.isfinite(ref) & (ref != 0)
        assert np.array_equal(np.isnan(got), np.isnan(ref)), name
        # 4e-15 measured for every model except CMOD-IFR2 near the zeros of its azimuth modulation (9e-12: cancellation)
        assert np.max(np.abs(got[ok] - ref[ok]) / np.abs(ref[ok])) < (1e-10 if name == "gmf_cmodifr2" else 1e-13), name
    # model layer: a large broadcast evaluation takes the device route, a small one the host route; same values
    m = windspeed.get_model("gmf_cmod5n")
    inc2 = np.broadcast_to(np.linspace(20, 45, 600), (500, 600)).copy()
    w2 = rng.uniform(1, 30, inc2.shape)
    p2 = rng.uniform(0, 360, inc2.shape)
    assert inc2.size >= options.gmf_device_min_size
    big = np.asarray(m(inc2, w2, p2))
    small = np.asarray(m(inc2[:2], w2[:2], p2[:2]))
    assert big.shape == inc2.shape and np.allclose(big[:2], small, rtol=1e-12, atol=0)
    assert np.allclose(big, ogmf.gmf_cmod5n(inc2, w2, p2), rtol=1e-12, atol=0)


def test_plain_c_caller(tmp_path):
    """The boundary is a flat C ABI: a C99 program (tests/abi_c/abi_smoke.c: no Python, no torch) linked against libxsw
    inverts a dual-pol problem on host buffers; indices equal the oracle's, winds agree to 1e-9 (the optional libm
    tables are left NULL there, so the last bit of cos/sin may differ from numpy's)."""
    import shutil
    import subprocess
    from oracle import gmf, lut as olut
    from oracle import invert as oinv
    from xsarsea_amd import _build
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    lib = _build.LIB
    exe = tmp_path / "abi_smoke"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I", os.path.join(REPO, "include"), "-o", str(exe),
                    os.path.join(REPO, "tests", "abi_c", "abi_smoke.c"), "-L", os.path.dirname(lib), "-lxsw",
                    "-Wl,-rpath," + os.path.dirname(lib)], check=True)
    rng = np.random.default_rng(11)
    inc_ax, w_ax, phi_ax = np.linspace(20, 45, 6), np.linspace(0.5, 40, 80), np.linspace(0, 180, 37)
    wcr_ax = np.linspace(3, 60, 58)
    co = 10 * np.log10(gmf.gmf_cmod5n(inc_ax[:, None, None], w_ax[None, :, None], phi_ax[None, None, :]) + 1e-15)
    cr = 10 * np.log10(gmf.GMFS["gmf_s1_v2"][0](inc_ax[:, None], wcr_ax[None, :]) + 1e-15)
    n = 333
    inc = rng.uniform(19, 46, n)
    wt, pt = rng.uniform(1, 35, n), rng.uniform(-180, 180, n)
    sco = oinv.to_db(gmf.gmf_cmod5n(inc, wt, pt) * rng.gamma(50, 1 / 50, n))
    scr = oinv.to_db(gmf.GMFS["gmf_s1_v2"][0](inc, np.maximum(wt, 3)) * rng.gamma(50, 1 / 50, n))
    dsig = 10 ** rng.uniform(-2, 0, n)
    anc = wt * np.exp(1j * np.deg2rad(pt)) + rng.normal(0, 2, n) + 1j * rng.normal(0, 2, n)
    sco[3], scr[4], inc[5], anc[6] = np.nan, np.nan, np.nan, complex(np.nan, 0)
    prob = tmp_path / "problem.bin"
    with open(prob, "wb") as f:
        np.array([len(inc_ax), len(w_ax), len(phi_ax), len(wcr_ax), n], dtype=np.int32).tofile(f)
        np.array([0.1]).tofile(f)
        for arr in (inc_ax, w_ax, phi_ax, co, wcr_ax, cr, inc, sco, scr, dsig):
            np.ascontiguousarray(arr, dtype=np.float64).tofile(f)
        np.ascontiguousarray(anc, dtype=np.complex128).view(np.float64).tofile(f)
    res = tmp_path / "result.bin"
    env = dict(os.environ)
    r = subprocess.run([str(exe), str(prob), str(res)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(res, dtype=np.uint8)
    out_co = raw[: 16 * n].view(np.complex128)
    out_cr = raw[16 * n: 32 * n].view(np.complex128)
    idx = raw[32 * n:].view(np.int32).reshape(n, 3)
    p = oinv.Prepared(olut.Lut(co, inc_ax, w_ax, phi_ax, "dB", "x", "co", "VV"), olut.Lut(cr, inc_ax, wcr_ax, None, "dB", "x", "cr", "VH"))
    o = oinv.invert_numpy(p, inc, sco, scr, dsig, anc, return_idx=True)
    assert np.array_equal(idx, o[2])
    assert_complex_close(out_co, o[0], rtol=1e-9, what="C caller co")
    assert_complex_close(out_cr, o[1], rtol=1e-9, what="C caller cr")


def test_threaded_callers_share_the_context_safely():
    """dask's threaded scheduler calls the reference's function from several threads; here they share one device
    context (not thread-safe at the C level), serialised by the Python layer -- also when they alternate between models
    (= LUT re-uploads).  Results must equal the sequential ones."""
    from concurrent.futures import ThreadPoolExecutor
    from xsarsea_amd import windspeed
    rng = np.random.default_rng(5)
    jobs = []
    for k in range(8):
        shape = (int(rng.integers(20, 60)), int(rng.integers(50, 200)))
        n = shape[0] * shape[1]
        inc = rng.uniform(20, 45, n).reshape(shape)
        wt, pt = rng.uniform(1, 30, n), rng.uniform(-180, 180, n)
        anc = (wt * np.exp(1j * np.deg2rad(pt)) + rng.normal(0, 2, n)).reshape(shape)
        model = "gmf_cmod5n" if k % 2 == 0 else "gmf_cmod5"
        s = (windspeed.get_model(model)(inc.ravel(), wt, pt, broadcast=True) * rng.gamma(50, 1 / 50, n)).reshape(shape)
        jobs.append((inc, s, anc, model))

    def run(job):
        inc, s, anc, model = job
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return windspeed.invert_from_model(inc, s, ancillary_wind=anc, model=model, resolution="low")

    seq = [run(j) for j in jobs]
    with ThreadPoolExecutor(max_workers=4) as ex:
        par = list(ex.map(run, jobs * 3))
    for i, r in enumerate(par):
        assert bits_equal(r, seq[i % len(jobs)]), f"job {i} differs when run from a thread pool"
