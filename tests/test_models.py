"""CPU: host logic of the drop-in layer -- model registry, GMFs, LUT preparation -- against the oracle
and the golden vectors.  (No device compute here.)"""
import warnings

import os

import numpy as np
import pytest

from conftest import golden
from oracle import gmf as ogmf
from oracle import lut as olut
from xsarsea_amd import windspeed
from xsarsea_amd.windspeed import cmod7, gmfs, lut as plut, models


def test_registry_names_and_aliases():
    df = windspeed.available_models()
    for name in ogmf.GMFS:
        assert name in df.index
    assert set(df.columns) == {"alias", "pol", "model"}
    assert windspeed.get_model("cmod5n") is windspeed.get_model("gmf_cmod5n")
    assert windspeed.get_model(windspeed.get_model("gmf_s1_v2")).name == "gmf_s1_v2"
    with pytest.raises(KeyError, match="not found"):
        windspeed.get_model("gmf_does_not_exist")
    vh = windspeed.available_models(pol="VH")
    assert len(vh) >= 8 and all(vh.pol == "VH")
    m = windspeed.get_model("gmf_cmod5n")
    assert m.iscopol and not m.iscrosspol and m.phi_range == [0.0, 180.0] and m.wspd_range == [0.2, 50.0]
    x = windspeed.get_model("gmf_s1_v2")
    assert x.iscrosspol and x.phi_range is None and x.wspd_range == [3.0, 80.0]


def test_register_decorator_contract():
    """The plugin decorator as the reference's test uses it (test_xsarsea.py:8-21)."""
    @gmfs.GmfModel.register(inc_range=[17., 50.], wspd_range=[3., 80.], pol="VH", units="linear", defer=False)
    def gmf_dummy(inc, wspd, phi=None):
        a = 0.00013106836021008122 + -4.530598283705591e-06 * inc + 4.429277425062766e-08 * inc ** 2
        b = 1.3925444179360706 + 0.004157838450541205 * inc + 3.4735809771069953e-05 * inc ** 2
        return a * wspd ** b

    m = windspeed.get_model("gmf_dummy")
    assert m.inc_range == [17., 50.] and m.pol == "VH"
    v = np.asarray(m(np.arange(20, 22), np.arange(10, 12)))
    assert np.allclose(v, [[0.00179606, 0.00207004], [0.0017344, 0.00200004]], rtol=2e-6)  # gmfs.py:60-63
    assert np.isscalar(m(35, 15, 90))
    with pytest.raises(ValueError, match="must start with"):
        gmfs.GmfModel.register(pol="VV")(lambda inc, wspd, phi: 1.0)

    @gmfs.GmfModel.register("gmf_scalar_only", pol="VV", units="linear", defer=False)
    def _scalar_only(inc, wspd, phi):  # branches on a scalar: numpy arrays make `if` ambiguous
        import math
        mod = 1 + 0.3 * math.cos(math.radians(phi))
        base = 0.01 * wspd * mod if wspd > 2 else 0.001 * mod
        return base * (40.0 / inc)

    ms = windspeed.get_model("gmf_scalar_only")
    assert ms.phi_range == [0.0, 180.0]
    lut = ms._lut(units="linear", resolution="low", inc_step_lr=10.0, wspd_step_lr=10.0, phi_step_lr=45.0)
    assert lut.shape == (6, 6, 5) and np.isfinite(lut.values).all()


@pytest.mark.parametrize("name", sorted(ogmf.GMFS))
def test_product_gmfs_match_reference_lattice(name):
    g = golden("gmf_lattice.npz")
    f = windspeed.get_model(name)._gmf_pyfunc_scalar
    with np.errstate(all="ignore"):
        v = np.broadcast_to(f(g["inc"][:, None, None], g["wspd"][None, :, None], g["phi"][None, None, :]), g[name].shape)
    ok = np.isfinite(g[name]) & (g[name] != 0)
    assert np.array_equal(np.isnan(v), np.isnan(g[name]))
    assert np.max(np.abs(v[ok] - g[name][ok]) / np.abs(g[name][ok])) < 1e-13


def test_lut_policy_and_values_lowres():
    """resolution='low' -> raw grids, no interpolation; identical to the oracle's restatement."""
    for name in ("gmf_cmod5n", "gmf_s1_v2"):
        mine = windspeed.get_model(name)._lut(units="dB", resolution="low")
        ref = olut.to_lut(name, resolution="low")
        assert mine.attrs["units"] == "dB" and mine.attrs["resolution"] == "low"
        assert mine.shape == ref.values.shape and np.array_equal(mine.values, ref.values)
        assert np.array_equal(mine.wspd, ref.wspd) and np.array_equal(mine.incidence, ref.incidence)
    co = windspeed.get_model("gmf_cmod5n")._lut(units="dB", resolution="low")
    assert co.shape == (51, 250, 73) and co.dims == ("incidence", "wspd", "phi")


def test_lut_default_highres_equals_oracle(default_luts):
    """Default call (raw low-res -> linear interpolation inc, wspd, phi -> dB) == oracle, bit for bit."""
    lco, lcr = default_luts
    mine = windspeed.get_model("gmf_cmod5n")._lut(units="dB")
    assert mine.shape == (501, 499, 181) and mine.attrs["resolution"] == "high"
    assert np.array_equal(mine.values, lco.values)
    cr = windspeed.get_model("gmf_s1_v2")._lut(units="dB")
    assert cr.shape == (501, 771) and np.array_equal(cr.values, lcr.values)
    assert windspeed.get_model("gmf_cmod5n")._lut(units="dB") is mine  # memoised


def test_lerp_matches_scipy_bitwise():
    from scipy.interpolate import interp1d
    rng = np.random.default_rng(0)
    y = rng.standard_normal((7, 11, 5))
    x_old = np.linspace(0.2, 50, 11)
    x_new = np.linspace(0.2, 50, 101)
    ref = interp1d(x_old, y, kind="linear", axis=1, bounds_error=True)(x_new)
    assert np.array_equal(plut.lerp_axis(y, x_old, x_new, 1), ref)
    with pytest.raises(ValueError):
        plut.lerp_axis(y, x_old, np.array([0.0, 1.0]), 1)


def test_lut_validation_errors():
    """Same exception types as Model._normalize_lut (models.py:84-105)."""
    class Fake:
        def __init__(self, dims, attrs):
            self.dims, self.attrs = dims, attrs
            self._v = np.zeros((2, 2))

        def __array__(self, dtype=None, copy=None):
            return self._v

        def __getitem__(self, k):
            return np.array([0.0, 1.0])

    with pytest.raises(KeyError):
        plut.Lut.from_any(Fake(("incidence", "wspd"), {"resolution": "high"}))
    with pytest.raises(ValueError, match="Unknown lut units"):
        plut.Lut.from_any(Fake(("incidence", "wspd"), {"units": "furlongs", "resolution": "high"}))
    with pytest.raises(IndexError):
        plut.Lut.from_any(Fake(("wspd", "incidence"), {"units": "dB", "resolution": "high"}))
    with pytest.raises(AssertionError):
        plut.Lut.from_any(Fake(("incidence", "wspd"), {"units": "dB"}))


def test_cmod7_binary_roundtrip(tmp_path):
    """CMOD7 file format (cmod7.py:27-40): float32 LE, one marker word each end, Fortran (250,73,51)."""
    rng = np.random.default_rng(1)
    table = rng.uniform(1e-4, 1.0, (250, 73, 51)).astype(np.float32)
    cmod7.write_cmod7_table(tmp_path / "gmf_cmod7_vv.dat_little_endian", table)
    m = cmod7.register_cmod7(str(tmp_path))
    assert m.name == "gmf_cmod7" and m.pol == "VV" and windspeed.get_model("cmod7") is m
    raw = m._raw_lut()
    assert raw.shape == (51, 250, 73) and raw.attrs["units"] == "linear" and raw.attrs["resolution"] == "low"
    assert np.array_equal(raw.values, np.transpose(table, (2, 0, 1)).astype(np.float64))
    hi = m._lut(units="dB")
    assert hi.shape == (501, 499, 181)
    # low-res grid points (inc every 10th, wspd 0.2*k -> index 2k-2... checked on inc & phi axes) are reproduced
    assert np.allclose(hi.values[::10, 0, ::5], 10 * np.log10(raw.values[:, 0, ::2] + 1e-15), rtol=0, atol=1e-9)
    with pytest.raises(FileNotFoundError):
        cmod7.register_cmod7(str(tmp_path / "nope"))


def test_invert_host_side_errors():
    """Argument routing errors raised before any device work (windspeed.py:88-120)."""
    inc = np.full((4, 4), 30.0)
    s = np.full((4, 4), 0.05)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(AssertionError):  # co-pol inversion without any valid ancillary wind (:107)
            windspeed.invert_from_model(inc, s, model="gmf_cmod5n", resolution="low")
        with pytest.raises(KeyError):
            windspeed.invert_from_model(inc, s, model="gmf_nope")
    with pytest.warns(UserWarning, match="Unable to check sigma0 pol"):
        try:
            windspeed.invert_from_model(inc, s, ancillary_wind=s + 0j, model="gmf_cmod5n", resolution="low")
        except Exception:
            pass  # no GPU here: the device call raises after the host-side warning


def test_utils_formulas():
    from xsarsea_amd.windspeed import utils
    s, n = np.array([1e-3, 2e-3]), np.array([1e-4, 1e-4])
    assert np.allclose(utils.get_dsig("nc_lut_cmodms1ahw", None, s, n), (1.25 / (s / n)) ** 4)
    assert np.allclose(utils.get_dsig("gmf_rs2_v2", None, s, n), 1 / np.sqrt((s / n) ** 8))
    with pytest.raises(ValueError):
        utils.get_dsig("other", None, s, n)
    inc = np.broadcast_to(np.linspace(20, 45, 50), (6, 50)).copy()
    noise = 10 ** ((-0.1 * inc - 20) / 10)
    noise[2, 5] = np.nan
    flat = utils.nesz_flattening(noise, inc)
    assert np.allclose(flat, 10 ** ((-0.1 * inc - 21) / 10), rtol=1e-9)
    with pytest.raises(IndexError):
        utils.nesz_flattening(noise[0], inc[0])


def test_direction_helpers():
    import xsarsea_amd as xa
    assert xa.dir_meteo_to_oceano(10) == 190 and xa.dir_oceano_to_meteo(190) == 10
    assert xa.dir_to_180(270) == -90 and xa.dir_to_360(-90) == 270
    assert np.isclose(xa.dir_meteo_to_sample(90.0, 0.0), 0.0)
    assert xa.dir_sample_to_meteo(0.0, 10.0) == 100.0


def test_lut_kwargs_do_not_leak_between_calls():
    """A call with step overrides must not change what a later default call builds (the reference mutates the
    model's steps in _raw_lut, gmfs.py:370-379; the build carries the generated steps with the LUT instead)."""
    m = windspeed.get_model("gmf_cmod5")
    before = (m.inc_step_lr, m.wspd_step_lr, m.phi_step_lr)
    a = m._lut(units="linear", resolution="low")
    b = m._lut(units="linear", resolution="low", inc_step_lr=2.0)
    assert b.shape[0] == 26 and a.shape[0] == 51
    assert (m.inc_step_lr, m.wspd_step_lr, m.phi_step_lr) == before
    m._lut_cache.clear()
    c = m._lut(units="linear", resolution="low")
    assert np.array_equal(a.values, c.values)


def test_pickle_lut_directories(tmp_path):
    """sarwing LUT layout (pickle_luts.py:26-73): sigma.npy transposed w.r.t. the pickled axes, dB, high resolution."""
    import pickle
    from xsarsea_amd.windspeed import pickle_luts
    rng = np.random.default_rng(2)
    inc, wspd, phi = np.arange(17.0, 50.05, 0.5), np.arange(0.5, 30.05, 0.5), np.arange(0.0, 180.5, 5.0)
    co = rng.uniform(-30, -5, (len(inc), len(wspd), len(phi)))   # (incidence, wspd, phi) is what to_lut must return
    d = tmp_path / "GMF_testco"
    d.mkdir()
    np.save(d / "sigma.npy", np.transpose(np.transpose(co, (1, 2, 0))))  # file = transpose of (wspd, phi, incidence)
    pickle.dump(inc, open(d / "incidence_angle.pkl", "wb"))
    pickle.dump((phi, wspd), open(d / "wind_speed_and_direction.pkl", "wb"))
    cr = rng.uniform(-35, -15, (len(inc), len(wspd)))
    d2 = tmp_path / "GMF_testcr"
    d2.mkdir()
    np.save(d2 / "sigma.npy", np.transpose(np.transpose(cr, (1, 0))))
    pickle.dump(inc, open(d2 / "incidence_angle.pkl", "wb"))
    pickle.dump(wspd, open(d2 / "wind_speed.pkl", "wb"))
    pickle_luts.register_pickle_luts(str(tmp_path))
    mco, mcr = windspeed.get_model("sarwing_lut__testco"), windspeed.get_model("sarwing_lut__testcr")
    assert mco.pol == "VV" and mcr.pol == "VH" and windspeed.get_model("testco") is mco
    lco = mco._lut(units="dB")
    assert lco.dims == ("incidence", "wspd", "phi") and np.array_equal(lco.values, co)
    assert mco.phi_range == [0.0, 180.0] and mco.wspd_step == 0.5 and mco.inc_range == [17.0, 50.0]
    lcr = mcr._lut(units="dB")
    assert np.array_equal(lcr.values, cr) and lcr.phi is None
    lin = mcr._lut(units="linear")
    assert np.allclose(lin.values, 10 ** (cr / 10))


def test_netcdf_lut_roundtrip(tmp_path):
    """`Model.to_netcdf` -> `register_nc_luts` -> `NcLutModel` (models.py:232-262, :350-450) without xarray: classic netCDF-3
    through scipy, the reference's schema (sigma0_model over incidence/wspd[/phi], global attrs units, resolution, model, pol,
    *_range, *_step).  Co-pol models are stored at low resolution in dB, cross-pol at high resolution; reading back gives the
    same table, axes and metadata, and the model then prepares a default-resolution dB LUT like any other."""
    from scipy.io import netcdf_file
    from xsarsea_amd.windspeed import nc_io
    if models.xr is not None:
        pytest.skip("xarray present: the scipy route is not the one taken")
    co, cr = windspeed.get_model("gmf_cmod5n"), windspeed.get_model("gmf_s1_v2")
    p_co, p_cr = tmp_path / "nc_lut_rt_cmod5n.nc", tmp_path / "nc_lut_rt_s1_v2.nc"
    co.to_netcdf(str(p_co))
    cr.to_netcdf(str(p_cr))
    assert nc_io.is_classic_netcdf(str(p_co))
    with netcdf_file(str(p_co), "r", mmap=False) as f:
        assert f.variables["sigma0_model"].dimensions == ("incidence", "wspd", "phi") and f.variables["sigma0_model"].shape == (51, 250, 73)
        assert f.units == b"dB" and f.resolution == b"low" and f.model == b"cmod5n" and f.pol == b"VV"
        assert list(f.inc_range) == [16.0, 66.0] and list(f.phi_range) == [0.0, 180.0] and float(f.wspd_step) == 0.2
    windspeed.register_nc_luts(str(tmp_path))
    avail = windspeed.available_models()
    assert "nc_lut_rt_cmod5n" in avail.index and "nc_lut_rt_s1_v2" in avail.index
    m_co, m_cr = windspeed.get_model("nc_lut_rt_cmod5n"), windspeed.get_model("nc_lut_rt_s1_v2")
    assert isinstance(m_co, models.NcLutModel) and m_co.pol == "VV" and m_co.iscopol and m_cr.iscrosspol and m_co.short_name == "cmod5n"
    assert m_co.inc_step_lr == 1.0 and m_co.wspd_step_lr == 0.2 and m_co.phi_step_lr == 2.5 and m_co.units == "dB"
    raw = m_co._raw_lut()
    ref = co._lut(units="dB", resolution="low")
    assert np.array_equal(raw.values, ref.values) and np.array_equal(raw.wspd, ref.wspd) and raw.attrs["resolution"] == "low"
    raw_cr = m_cr._raw_lut()
    assert np.array_equal(raw_cr.values, cr._lut(units="dB", resolution="high").values) and raw_cr.phi is None
    hi = m_co._lut(units="dB")  # low -> high in the LUT's own (dB) units, as the reference does for a dB table
    assert hi.shape == (501, 499, 181) and hi.attrs["units"] == "dB"
    assert np.allclose(hi.values[::10, ::2, ::5], raw.values[:, :, ::2], rtol=0, atol=1e-9)  # the shared grid nodes are kept
    assert m_cr._lut(units="dB").shape == (501, 771)
    # a truncated HDF5 file, and a file that is neither container, fail with a clear message
    bad = tmp_path / "nc_lut_hdf5.nc"
    bad.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    with pytest.raises((ValueError, NotImplementedError)):
        models.NcLutModel(str(bad))
    junk = tmp_path / "nc_lut_junk.nc"
    junk.write_bytes(b"not a netcdf file at all")
    with pytest.raises(ImportError, match="neither"):
        models.NcLutModel(str(junk))
    for n in ("nc_lut_rt_cmod5n", "nc_lut_rt_s1_v2"):
        models.Model._available_models.pop(n, None)


NC4 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nc4")
NC4_FILES = ["nc_lut_netcdf4_contiguous.nc", "nc_lut_netcdf4_deflate.nc", "nc_lut_netcdf4_deflate_fletcher_f32.nc", "nc_lut_netcdf4_crosspol.nc",
             "nc_lut_netcdf4_v18.nc", "nc_lut_netcdf4_many_attrs.nc", "nc_lut_h5netcdf_earliest.nc", "nc_lut_h5netcdf_deflate.nc",
             "nc_lut_h5netcdf_latest.nc"]


@pytest.mark.parametrize("fname", NC4_FILES)
def test_netcdf4_hdf5_lut_files(fname):
    """netCDF-4 (HDF5) LUT files in the layouts of both xarray backends -- netCDF4-python (creation order tracked: dense
    attribute storage in a fractal heap + v2 B-tree, fixed-length text attributes, 1-element numeric attributes, `_NCProperties`,
    `_FillValue`) and h5netcdf (variable-length strings in the global heap, old-style groups or the latest file format) --
    contiguous or chunked + shuffle + deflate (+ fletcher32), float64 or float32: the package's own HDF5 reader returns the
    table, the axes (in DIMENSION_LIST order) and the global attributes the fixture generator wrote (tests/golden/make_nc4_fixtures.py,
    h5py in the build container's conda), and `NcLutModel` builds the model from them like from a classic file."""
    from xsarsea_amd.windspeed import hdf5_min, models, nc_io
    path = os.path.join(NC4, fname)
    exp = np.load(path + ".expected.npz")
    assert hdf5_min.is_hdf5(path) and not nc_io.is_classic_netcdf(path)
    f = hdf5_min.File(path)
    has_phi = "phi" in exp
    assert f.names() == sorted(["incidence", "wspd", "sigma0_model"] + (["phi"] if has_phi else []))
    assert f.dims("sigma0_model") == (("incidence", "wspd", "phi") if has_phi else ("incidence", "wspd"))
    assert np.array_equal(np.asarray(f.read("sigma0_model"), np.float64), exp["values"])
    lut = nc_io.read_lut(path)
    assert np.array_equal(lut.values, exp["values"]) and lut.values.dtype == np.float64
    assert np.array_equal(lut.incidence, exp["incidence"]) and np.array_equal(lut.wspd, exp["wspd"])
    assert (lut.phi is None) == (not has_phi) and (not has_phi or np.array_equal(lut.phi, exp["phi"]))
    attrs = nc_io.read_attrs(path)
    for k in ("units", "resolution", "model", "pol"):
        assert attrs[k] == str(exp[k]), (k, attrs[k])
    for k in ("inc_range", "wspd_range") + (("phi_range",) if has_phi else ()):
        assert np.array_equal(np.asarray(attrs[k], np.float64), exp[k])
    for k in ("inc_step", "wspd_step") + (("phi_step",) if has_phi else ()):
        assert float(attrs[k]) == float(exp[k])
    assert not any(k.startswith("_NC") for k in attrs)
    if fname == "nc_lut_netcdf4_many_attrs.nc":  # 29 attributes: indirect root block of the heap, B-tree of depth 1
        assert sum(k.startswith("history_") for k in attrs) == 18 and attrs["history_03"].startswith("step 3: processing note")
    m = models.NcLutModel(path)
    try:
        assert m.pol == str(exp["pol"]) and m.short_name == "cmod_fixture" and m.iscopol == has_phi
        raw = m._raw_lut()
        assert np.array_equal(raw.values, exp["values"]) and raw.attrs["units"] == "dB" and raw.attrs["resolution"] == "low"
    finally:
        models.Model._available_models.pop(m.name, None)


def test_netcdf4_layout_v4_chunk_index_is_refused_clearly():
    """A chunked variable written with libver >= 1.10 bounds (HDF5 'version 4' data layout: fixed-array chunk index) is outside
    the reader's subset -- netCDF-4 writers keep the 1.8-compatible layout -- and says so; its attributes still read."""
    from xsarsea_amd.windspeed import nc_io
    path = os.path.join(NC4, "nc_lut_h5py_latest_chunked.nc")
    assert nc_io.read_attrs(path)["units"] == "dB"
    with pytest.raises(NotImplementedError, match="layout version 4"):
        nc_io.read_lut(path)


def test_hdf5_unwritten_storage_reads_as_the_fill_value():
    """chunks that were never written (and a contiguous dataset without an address) read as the dataset's fill value --
    the fill value message, as HDF5 / h5py return it (fixture written by h5py: tests/golden/nc4/unwritten_storage.h5) -- not as
    zeros; where the file defines none, HDF5's zeros come back but the reader says so (`undefined_fill_used`), and `read_lut`
    refuses such a table."""
    from xsarsea_amd.windspeed import hdf5_min
    path = os.path.join(NC4, "unwritten_storage.h5")
    exp = np.load(path + ".expected.npz")
    f = hdf5_min.File(path)
    for name in ("partly", "never"):
        assert np.array_equal(f.read(name), exp[name]), name
    assert f.read("partly")[5, 7] == -999.0 and f.read("never")[0, 0] == 7.5 and not f.undefined_fill_used
    assert np.array_equal(f.read("partly_nofill"), exp["partly_nofill"]) and f.undefined_fill_used
    assert np.array_equal(f.read("never_int"), exp["never_int"])
    with pytest.raises(KeyError):
        f.read("no_such_dataset")


def test_hdf5_fletcher32_is_verified():
    """A flipped byte inside a Fletcher-32-protected chunk is reported (ValueError naming the file), not decoded."""
    import shutil
    import tempfile
    from xsarsea_amd.windspeed import hdf5_min
    src = os.path.join(NC4, "nc_lut_netcdf4_deflate_fletcher_f32.nc")
    data = bytearray(open(src, "rb").read())
    f = hdf5_min.File(src)
    assert hdf5_min.fletcher32(b"") == 0 and hdf5_min.fletcher32(b"\x01\x02\x03") == ((0x0102 + 0x0102 + 0x0300) << 16 | (0x0102 + 0x0300))
    # find a chunk through the reader's own B-tree walk: the first leaf's address
    obj = f._dataset("sigma0_model")
    lo, _ = f._msg(obj, 0x08)
    btree = f.buf.u(lo + 3, 8)
    level = f.buf.d[btree + 5]
    assert level == 0
    rank = f.buf.d[lo + 2] - 1
    ksize = 8 + 8 * (rank + 1)
    addr = f.buf.u(btree + 8 + 16 + ksize, 8)
    data[addr + 5] ^= 0x40
    with tempfile.TemporaryDirectory() as td:
        bad = os.path.join(td, "flipped.nc")
        with open(bad, "wb") as out:
            out.write(data)
        with pytest.raises(ValueError, match="flipped.nc"):
            hdf5_min.File(bad).read("sigma0_model")


@pytest.mark.parametrize("fname", sorted(n for n in os.listdir(NC4) if n.endswith((".nc", ".h5"))))
def test_hdf5_reader_on_damaged_files(fname):
    """bit-flipped, truncated and partly zeroed copies of every fixture either still read or raise ValueError /
    NotImplementedError naming the problem -- never an IndexError / TypeError / struct.error / zlib.error from the middle of the
    parser, and never a hang (5 s alarm per case).  40 damaged copies per fixture, seeded."""
    import signal
    import tempfile
    from xsarsea_amd.windspeed import hdf5_min, nc_io
    src = open(os.path.join(NC4, fname), "rb").read()
    import zlib
    rng = np.random.default_rng(zlib.crc32(fname.encode()))

    def on_alarm(signum, frame):
        raise TimeoutError("hdf5_min hung on a damaged file")

    old = signal.signal(signal.SIGALRM, on_alarm)
    try:
        with tempfile.TemporaryDirectory() as td:
            for case in range(40):
                data = bytearray(src)
                kind = case % 4
                if kind == 0:  # a few flipped bits, mostly in the metadata-heavy first 4 KB
                    for _ in range(int(rng.integers(1, 6))):
                        pos = int(rng.integers(0, min(len(data), 4096))) if rng.random() < 0.7 else int(rng.integers(0, len(data)))
                        data[pos] ^= 1 << int(rng.integers(0, 8))
                elif kind == 1:  # truncated
                    data = data[: int(rng.integers(16, len(data)))]
                elif kind == 2:  # a zeroed span
                    a = int(rng.integers(0, len(data) - 8))
                    data[a:a + int(rng.integers(8, 512))] = bytes(min(512, len(data) - a))[: len(data[a:a + int(rng.integers(8, 512))])]
                else:  # random bytes over a span
                    a = int(rng.integers(0, len(data) - 8))
                    n = int(rng.integers(4, 64))
                    data[a:a + n] = rng.integers(0, 256, len(data[a:a + n]), dtype=np.uint8).tobytes()
                path = os.path.join(td, f"damaged_{case}.nc")
                with open(path, "wb") as out:
                    out.write(data)
                signal.alarm(5)
                try:
                    f = hdf5_min.File(path)
                    for name in f.names():
                        try:
                            f.read(name)
                            f.dims(name)
                        except (ValueError, NotImplementedError, KeyError):
                            pass
                    if fname.endswith(".nc"):
                        nc_io.read_lut(path)
                except (ValueError, NotImplementedError, KeyError, ImportError, IndexError) as e:
                    # IndexError: only nc_io's own "Bad dims" message (the reference's: models.py:84-105), never the parser's
                    assert not isinstance(e, IndexError) or "Bad dims" in str(e), (fname, case, repr(e))
                    assert not isinstance(e, KeyError) or "no " in str(e) or "units" in str(e) or "resolution" in str(e), (fname, case, repr(e))
                finally:
                    signal.alarm(0)
    finally:
        signal.signal(signal.SIGALRM, old)
