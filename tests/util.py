"""Shared helpers of the parity tests (oracle side + comparison rules)."""
import numpy as np

from oracle import invert as oinv
from oracle import lut as olut


def small_luts(d):
    lco = olut.Lut(d["lut_co"], d["lut_inc"], d["lut_wspd"], d["lut_phi"], "dB", "x", "co", "VV")
    lcr = olut.Lut(d["lut_cr"], d["lut_inc"], d["lut_wspd_cr"], None, "dB", "x", "cr", "VH")
    return lco, lcr


def lut_dicts(lco, lcr):
    co = cr = None
    if lco is not None:
        co = dict(db=lco.values, inc=lco.incidence, wspd=lco.wspd, phi=lco.phi, **host_tables(lco.wspd, lco.phi))
    if lcr is not None:
        cr = dict(db=lcr.values, inc=lcr.incidence, wspd=lcr.wspd)
    return co, cr


def host_tables(wspd, phi):
    """The platform-dependent transcendental tables of xsw_lut, evaluated with numpy by the very
    expressions of the reference (windspeed.py:167-168, :235-236, :257, :270-276)."""
    wspd = np.asarray(wspd, dtype=np.float64)
    phi = np.asarray(phi, dtype=np.float64)
    e = np.stack([np.exp(1j * np.deg2rad(phi)), np.exp(1j * np.deg2rad(-phi))])          # (2, n_phi)
    sol = wspd[None, :, None] * e[:, None, :]                                             # (2, n_w, n_phi)
    unit = np.exp(1j * np.angle(sol))
    return dict(cos_phi=np.cos(np.radians(phi)), sin_phi=np.sin(np.radians(phi)),
                out_dir=np.stack([e.real, e.imag], axis=-1), abs_co=np.abs(sol[0]),
                dual_dir=np.stack([unit.real, unit.imag], axis=-1))


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    va = a.view(np.float64) if a.dtype == np.complex128 else a
    vb = b.view(np.float64) if b.dtype == np.complex128 else b
    return np.array_equal(va, vb, equal_nan=True)


def assert_complex_close(got, ref, rtol=1e-12, what=""):
    """NaN patterns identical (separately on Re and Im), finite values within rtol of |ref|."""
    got = np.asarray(got, dtype=np.complex128)
    ref = np.asarray(ref, dtype=np.complex128)
    assert got.shape == ref.shape, what
    for part in (np.real, np.imag):
        g, r = part(got), part(ref)
        assert np.array_equal(np.isnan(g), np.isnan(r)), f"{what}: NaN pattern differs"
    ok = ~np.isnan(ref.real) & ~np.isnan(ref.imag)
    if ok.any():
        err = np.abs(got[ok] - ref[ok])
        scale = np.maximum(np.abs(ref[ok]), 1e-3)
        assert np.max(err / scale) <= rtol, f"{what}: max rel err {np.max(err / scale)}"


def oracle_full(inc, s_vv, s_vh, dsig_cr, anc, lco, lcr, fast_c=True):
    """Oracle (co, cr_raw, idx) for the dual call; uses the C restatement (incidence-major copy) when
    fast_c, else the numpy restatement."""
    p = oinv.Prepared(lco, lcr)
    nan = np.full(np.shape(inc), np.nan)
    s_co_db = oinv.to_db(s_vv) if s_vv is not None else nan
    s_cr_db = oinv.to_db(s_vh) if s_vh is not None else nan
    if dsig_cr is None:
        dsig_cr = nan
    elif np.isscalar(dsig_cr):
        dsig_cr = s_vh * 0 + dsig_cr
    if anc is None:
        anc = nan
    if fast_c:
        from oracle import cport
        return cport.invert_numpy(p, inc, s_co_db, s_cr_db, dsig_cr, anc, return_idx=True, reference_layout=False)
    return oinv.invert_numpy(p, inc, s_co_db, s_cr_db, dsig_cr, anc, return_idx=True)
