#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by EXECUTING THE REFERENCE'S OWN CODE.

Runs only in the build container (needs /root/reference); nothing here runs on the GPU box and no
reference text is copied: only numeric inputs/outputs are written (`*.npz`).

How the reference is executed.  `import xsarsea` fails in this image with an
ordinary `ModuleNotFoundError` (xarray, numba, dask are not installed), so the reference's source
FILES are loaded by path under their real dotted names, with two tiny in-memory stand-ins for the
absent third-party modules:
  * `xarray`: `zeros_like` raises TypeError, which is exactly what steers the reference into its
    own pure-numpy branch (windspeed.py:381-386); `where` = numpy.where.
  * `numba`: `guvectorize` returns a wrapper that honours the gufunc contract of
    windspeed.py:306-323 (cast to float64/complex128, flatten, allocate two complex128 outputs,
    call the decorated function once).  The reference's kernel body `__invert_from_model_1d`
    (windspeed.py:183-282) therefore runs UNMODIFIED on numpy inputs.
The GMF scalar functions (gmfs_impl.py) are captured through a recording `GmfModel.register`.

What is NOT executed from the reference (needs real xarray): `Model._normalize_lut`'s
`DataArray.interp`.  LUTs fed to the reference kernel are therefore built by `oracle.lut` (raw GMF
grid from the reference's scalar GMF is cross-checked against it in gmf_lattice.npz) or are small
explicit arrays stored inside the fixture.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz, ~2 min)
        python tests/golden/make_golden.py phi90_f64  (only the named kernel_small_* fixtures)
        python tests/golden/make_golden.py crosspol_prep  (only crosspol_prep.npz: the reference's windspeed/utils.py)
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src/xsarsea"
sys.path.insert(0, REPO)


# ----------------------------------------------------------------------------- stand-ins
def _install_stubs():
    xr = types.ModuleType("xarray")

    class DataArray:  # only referenced by isinstance checks
        pass

    def zeros_like(*a, **k):
        raise TypeError("stub xarray: not a DataArray")

    xr.DataArray = DataArray
    xr.zeros_like = zeros_like
    xr.where = np.where
    sys.modules["xarray"] = xr

    nb = types.ModuleType("numba")

    class _T:
        def __getitem__(self, k):
            return self

        def __call__(self, *a):
            return self

    for n in ("float64", "float32", "complex128", "void"):
        setattr(nb, n, _T())

    def _ident(*a, **k):
        def deco(f):
            return f
        return deco

    nb.njit = _ident
    nb.vectorize = _ident

    def guvectorize(sigs, layout, **kw):
        assert layout == "(n),(n),(n),(n),(n)->(n),(n)", layout

        def deco(f):
            def w(*arrays):
                shape = np.shape(arrays[0])
                types_ = (np.float64,) * 4 + (np.complex128,)
                flat = [
                    np.ascontiguousarray(np.broadcast_to(np.asarray(a), shape)).astype(t).ravel()
                    for a, t in zip(arrays, types_)
                ]
                out_co = np.empty(flat[0].size, dtype=np.complex128)
                out_cr = np.empty(flat[0].size, dtype=np.complex128)
                f(*flat, out_co, out_cr)
                return out_co.reshape(shape), out_cr.reshape(shape)
            return w
        return deco

    nb.guvectorize = guvectorize
    sys.modules["numba"] = nb

    for pkg, path in (("xsarsea", REF), ("xsarsea.windspeed", REF + "/windspeed")):
        m = types.ModuleType(pkg)
        m.__path__ = [path]
        sys.modules[pkg] = m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    _install_stubs()
    _load("xsarsea.utils", REF + "/utils.py")
    _load("xsarsea.windspeed.utils", REF + "/windspeed/utils.py")
    models = _load("xsarsea.windspeed.models", REF + "/windspeed/models.py")
    ws = _load("xsarsea.windspeed.windspeed", REF + "/windspeed/windspeed.py")

    # GMF scalars: recording GmfModel.register
    captured = {}
    fake = types.ModuleType("xsarsea.windspeed.gmfs")

    class GmfModel:
        @classmethod
        def register(cls, name=None, pol=None, units="linear", defer=True, **kw):
            def deco(f):
                captured[name or f.__name__] = f
                return f
            return deco

    fake.GmfModel = GmfModel
    sys.modules["xsarsea.windspeed.gmfs"] = fake
    _load("xsarsea.windspeed.gmfs_impl", REF + "/windspeed/gmfs_impl.py")
    return models, ws, captured


# ----------------------------------------------------------------------------- model adaptor
class _DuckLut:
    """What windspeed.py:144-150 / :171-176 touch on a LUT: transpose(*dims), .wspd/.phi/.incidence."""

    def __init__(self, values, incidence, wspd, phi):
        self._v = values
        self.incidence, self.wspd, self.phi = incidence, wspd, phi
        self._dims = ("incidence", "wspd", "phi") if phi is not None else ("incidence", "wspd")

    def transpose(self, *dims):
        return np.transpose(self._v, [self._dims.index(d) for d in dims])


def make_model(models_mod, name, pol, values, incidence, wspd, phi):
    class _M(models_mod.Model):
        def __init__(self):
            self.name, self.pol = name, pol

        def to_lut(self, units="dB", **kw):
            assert units == "dB"
            return _DuckLut(values, incidence, wspd, phi)

        def _raw_lut(self):
            raise NotImplementedError

        short_name = None

    return _M()


# ----------------------------------------------------------------------------- inputs
def synth_small_lut(rng, n_inc=9, n_wspd=40, n_phi=19, phi_max=180.0, noise=0.3):
    """Small explicit co-pol + cross-pol dB LUTs (stored in the fixture: self-contained goldens)."""
    from oracle import gmf
    inc = np.linspace(20.0, 44.0, n_inc)
    wspd = np.linspace(0.5, 39.5, n_wspd)
    phi = np.linspace(0.0, phi_max, n_phi)
    co = gmf.gmf_cmod5n(inc[:, None, None], wspd[None, :, None], phi[None, None, :])
    co = 10 * np.log10(co + 1e-15) + noise * rng.standard_normal(co.shape)
    wspd_cr = np.linspace(3.0, 60.0, 58)
    cr = gmf.GMFS["gmf_s1_v2"][0](inc[:, None], wspd_cr[None, :])
    cr = 10 * np.log10(cr + 1e-15) + noise * rng.standard_normal(cr.shape)
    return dict(co=co, inc=inc, wspd=wspd, phi=phi, cr=cr, wspd_cr=wspd_cr)


def synth_pixels(rng, shape, inc_lo, inc_hi, dtype, with_edges=True):
    """Synthetic pixels around a CMOD5.N truth + the edge cases of SURVEY.md section 8c (G3)."""
    from oracle import gmf
    n = int(np.prod(shape))
    inc = rng.uniform(inc_lo, inc_hi, n)
    w_t = rng.uniform(1.0, 30.0, n)
    phi_t = rng.uniform(-180.0, 180.0, n)
    s_vv = gmf.gmf_cmod5n(inc, w_t, phi_t) * rng.gamma(100.0, 1 / 100.0, n)
    s_vh = gmf.GMFS["gmf_s1_v2"][0](inc, np.maximum(w_t, 3.0)) * rng.gamma(100.0, 1 / 100.0, n) + 10 ** -3.5
    anc = w_t * np.exp(1j * np.deg2rad(phi_t)) + rng.normal(0, 1.5, n) + 1j * rng.normal(0, 1.5, n)
    dsig_cr = (1.25 / (s_vh / 10 ** -3.5)) ** 4.0
    if with_edges and n >= 64:
        inc[0] = np.nan                      # NaN incidence -> NaN, NaN
        s_vv[1] = np.nan                     # NaN co sigma0 -> cross-only for that pixel
        anc[2] = np.nan + 0j                 # NaN ancillary with valid sigma0 -> NaN, NaN
        s_vv[3] = 0.0                        # -150 dB
        s_vv[4] = -1e-3                      # log10(negative) -> NaN dB
        inc[5] = inc_lo - 7.0                # below LUT range: clamps to first bin
        inc[6] = inc_hi + 9.0                # above LUT range: clamps to last bin
        anc[7] = complex(anc[7].real, 0.0)   # Im(anc) == 0
        anc[8] = complex(0.0, 0.0)           # zero ancillary
        dsig_cr[9] = np.nan                  # NaN dsig_cr -> no cross search
        s_vh[10] = np.nan                    # NaN cross sigma0
        anc[11] = complex(np.nan, 1.0)       # half-NaN ancillary
        w11 = 2.0
        s_vv[12] = gmf.gmf_cmod5n(inc[12], w11, 30.0)   # |co| < 5 select case
        anc[12] = w11 * np.exp(1j * np.deg2rad(30.0))
        s_vh[12] = gmf.GMFS["gmf_s1_v2"][0](inc[12], 3.0)
        anc[13] = complex(-abs(anc[13].real), -0.0)   # negative zero imaginary
        anc[14] = np.conj(anc[15])                    # mirrored pair
        s_vv[14] = s_vv[15]
        inc[14] = inc[15]
        s_vv[16] = np.inf
        anc[17] = 200.0 + 150.0j                      # ancillary far outside the grid
        anc[18] = complex(0.05, -0.02)                # tiny ancillary, negative Im
    if dtype == np.float32:
        return (inc.astype(np.float32).reshape(shape), s_vv.astype(np.float32).reshape(shape),
                s_vh.astype(np.float32).reshape(shape), dsig_cr.astype(np.float32).reshape(shape),
                anc.astype(np.complex64).reshape(shape))
    return (inc.reshape(shape), s_vv.reshape(shape), s_vh.reshape(shape), dsig_cr.reshape(shape),
            anc.reshape(shape))


def run_reference(ws, models_mod, luts, inc, s_vv, s_vh, dsig_cr, anc):
    """mono co-pol, dual-pol, cross-only (no ancillary) through the reference's invert_from_model."""
    import warnings
    m_co = make_model(models_mod, "gmf_golden_co", "VV", luts["co"], luts["inc"], luts["wspd"], luts["phi"])
    m_cr = make_model(models_mod, "gmf_golden_cr", "VH", luts["cr"], luts["inc"], luts["wspd_cr"], None)
    out = {}
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore")
        out["mono_co"] = ws.invert_from_model(inc, s_vv, ancillary_wind=anc, model=m_co)
        co, dual = ws.invert_from_model(inc, s_vv, s_vh, ancillary_wind=anc, dsig_cr=dsig_cr,
                                        model=(m_co, m_cr))
        out["dual_co"], out["dual_dual"] = co, dual
        out["cross_only"] = ws.invert_from_model(inc, s_vh, dsig_cr=0.1, model=m_cr)
    return out


def crosspol_prep_golden():
    """crosspol_prep.npz: inputs + outputs of the reference's windspeed/utils.py -- `get_dsig` (4 names), `get_dsig_wspd`
    (3 names), `nesz_flattening` (float64 and float32 rasters with NaN columns, NaN rows, scattered NaNs, a zero)."""
    import warnings
    ru = sys.modules["xsarsea.windspeed.utils"]
    rng = np.random.default_rng(20260320 + 30)
    out = {}
    # get_dsig
    n = 96
    inc = rng.uniform(17.0, 50.0, n)
    nesz = 10 ** rng.uniform(-3.8, -3.0, n)
    s_cr = nesz * 10 ** rng.uniform(-0.3, 2.0, n)
    s_cr[0], s_cr[1], nesz[2], inc[3] = 0.0, np.nan, np.nan, np.nan
    out.update(dsig_inc=inc, dsig_sigma0_cr=s_cr, dsig_nesz_cr=nesz)
    with np.errstate(all="ignore"):
        for name in ("gmf_s1_v2", "gmf_rs2_v2", "sarwing_lut_cmodms1ahw", "nc_lut_cmodms1ahw"):
            out["dsig_" + name] = ru.get_dsig(name, inc, s_cr, nesz)
            out["dsig32_" + name] = ru.get_dsig(name, inc.astype(np.float32), s_cr.astype(np.float32), nesz.astype(np.float32))
        # get_dsig_wspd
        U = np.concatenate([rng.uniform(0.0, 80.0, n - 4), [0.0, 30.0, 29.999, 80.0]])
        snr = np.concatenate([rng.uniform(-3.0, 20.0, n - 2), [0.0, np.nan]])
        out.update(dsigw_U=U, dsigw_SNR=snr)
        for name in ("dsig_wspd_rs2_v3", "dsig_wspd_s1_ew_rec_v3", "dsig_wspd_rcm_v3"):
            out[name] = ru.get_dsig_wspd(name, U, snr)
    # nesz_flattening
    L, S = 40, 96
    inc2 = np.linspace(29.0, 46.0, S)[None, :] + 0.02 * np.sin(np.arange(L) / 7.0)[:, None]
    noise = 10 ** ((-32.0 + 0.12 * (inc2 - 29.0) + 0.3 * rng.standard_normal((L, S))) / 10.0)
    noise[:, 5] = np.nan            # a column with no valid sample: the column mean is NaN there too
    noise[:, 50:53] = np.nan
    noise[7, :] = np.nan            # a whole line: filled from the column means
    noise[rng.random((L, S)) < 0.03] = np.nan
    noise[11, 20] = 0.0             # log10(0) = -inf: dropped from that line's fit
    noise[12, 21] = -1e-4           # log10(negative) = NaN: dropped
    inc2[3, 40] = np.nan
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out.update(nesz_noise=noise, nesz_inc=inc2, nesz_flat=ru.nesz_flattening(noise, inc2),
                   nesz_flat32=ru.nesz_flattening(noise.astype(np.float32), inc2.astype(np.float32)),
                   nesz_flat_allnan=ru.nesz_flattening(np.full((3, 8), np.nan), inc2[:3, :8]))
    np.savez_compressed(os.path.join(HERE, "crosspol_prep.npz"), **out)
    print("crosspol_prep.npz")


def main():
    models_mod, ws, gmfs = load_reference()
    from oracle import lut as olut

    if "crosspol_prep" in sys.argv[1:]:
        return crosspol_prep_golden()
    if not sys.argv[1:]:
        crosspol_prep_golden()
    only = set(sys.argv[1:])
    # ---- G1: scalar GMF lattice from the reference's own scalar functions
    inc_l = np.array([16.0, 20.3, 27.5, 35.0, 40.0, 47.7, 58.2, 66.0])
    wspd_l = np.array([0.2, 1.1, 3.0, 5.7, 10.0, 15.3, 24.9, 37.0, 50.0, 80.0])
    phi_l = np.array([0.0, 12.5, 45.0, 90.0, 133.0, 180.0, 225.0, 270.0, 359.0])
    lattice = {"inc": inc_l, "wspd": wspd_l, "phi": phi_l}
    with np.errstate(all="ignore"):
        for name, f in (gmfs.items() if not only else ()):
            vals = np.empty((len(inc_l), len(wspd_l), len(phi_l)))
            for i, a in enumerate(inc_l):
                for j, b in enumerate(wspd_l):
                    for k, c in enumerate(phi_l):
                        vals[i, j, k] = f(float(a), float(b), float(c))
            lattice[name] = vals
    # CMOD5.N branch boundaries (s ~ s0 and v2 ~ y0): dense wspd sweep at two incidences
    wd = np.linspace(0.2, 50.0, 499)
    for tag, a in ((("17", 17.0), ("60", 60.0)) if not only else ()):
        lattice["cmod5n_sweep_inc" + tag] = np.array([gmfs["gmf_cmod5n"](a, float(w), 37.0) for w in wd])
    lattice["sweep_wspd"] = wd
    if not only:
        np.savez_compressed(os.path.join(HERE, "gmf_lattice.npz"), **lattice)
        print("gmf_lattice.npz:", sorted(gmfs))

    # ---- G2: reference-scalar low-res CMOD5.N raw LUT sample + the same for gmf_s1_v2
    rng = np.random.default_rng(20260320)
    inc_lr = olut.grid([16.0, 66.0], 1.0)
    wspd_lr = olut.grid([0.2, 50.0], 0.2)
    phi_lr = olut.grid([0.0, 180.0], 2.5)
    ii = rng.integers(0, len(inc_lr), 2000)
    jj = rng.integers(0, len(wspd_lr), 2000)
    kk = rng.integers(0, len(phi_lr), 2000)
    if only:
        ii = ii[:0]
    samp = np.array([gmfs["gmf_cmod5n"](float(inc_lr[i]), float(wspd_lr[j]), float(phi_lr[k]))
                     for i, j, k in zip(ii, jj, kk)])
    wspd_cr_lr = olut.grid([3.0, 80.0], 0.2)
    jc = rng.integers(0, len(wspd_cr_lr), 2000)
    samp_cr = np.array([gmfs["gmf_s1_v2"](float(inc_lr[i]), float(wspd_cr_lr[j]), None)
                        for i, j in zip(ii, jc)])
    if not only:
        np.savez_compressed(os.path.join(HERE, "raw_lut_samples.npz"), ii=ii, jj=jj, kk=kk, cmod5n=samp,
                            jc=jc, s1_v2=samp_cr)
        print("raw_lut_samples.npz")

    # ---- G-small: self-contained kernel goldens (LUT arrays stored in the fixture)
    # phi90: the direction axis spans < 178 deg, the only way to reach the reference's phi_180 == False branch
    # (windspeed.py:152-156 makes every axis of >= 178 deg "symmetrical", a 0..360 one included)
    only = set(sys.argv[1:])
    for tag, phi_max, dtype in (("phi180_f64", 180.0, np.float64), ("phi360_f64", 360.0, np.float64),
                                ("phi180_f32", 180.0, np.float32), ("phi90_f64", 90.0, np.float64)):
        if only and tag not in only:
            continue
        rng = np.random.default_rng({"phi180_f64": 11, "phi360_f64": 12, "phi180_f32": 13, "phi90_f64": 14}[tag])
        luts = synth_small_lut(rng, n_phi={180.0: 19, 360.0: 37, 90.0: 10}[phi_max], phi_max=phi_max)
        inc, s_vv, s_vh, dsig_cr, anc = synth_pixels(rng, (24, 40), 18.0, 46.0, dtype)
        out = run_reference(ws, models_mod, luts, inc, s_vv, s_vh, dsig_cr, anc)
        np.savez_compressed(os.path.join(HERE, f"kernel_small_{tag}.npz"),
                            lut_co=luts["co"], lut_cr=luts["cr"], lut_inc=luts["inc"], lut_wspd=luts["wspd"],
                            lut_phi=luts["phi"], lut_wspd_cr=luts["wspd_cr"],
                            inc=inc, sigma0_vv=s_vv, sigma0_vh=s_vh, dsig_cr=dsig_cr, anc=anc, **out)
        print(f"kernel_small_{tag}.npz")

    if only:
        return

    # ---- G-default: default-resolution LUT (rebuilt at test time by oracle.lut.to_lut), 48x48 px
    lut_co = olut.to_lut("gmf_cmod5n")
    lut_cr = olut.to_lut("gmf_s1_v2")
    luts = dict(co=lut_co.values, inc=lut_co.incidence, wspd=lut_co.wspd, phi=lut_co.phi,
                cr=lut_cr.values, wspd_cr=lut_cr.wspd)
    fingerprint = dict(
        lut_co_sum=float(lut_co.values.sum()), lut_cr_sum=float(lut_cr.values.sum()),
        lut_co_samples=lut_co.values.ravel()[:: 45011].copy(), lut_cr_samples=lut_cr.values.ravel()[:: 397].copy(),
    )
    for tag, dtype in (("f64", np.float64), ("f32", np.float32)):
        rng = np.random.default_rng({"f64": 21, "f32": 22}[tag])
        inc, s_vv, s_vh, dsig_cr, anc = synth_pixels(rng, (48, 48), 17.0, 65.0, dtype)
        out = run_reference(ws, models_mod, luts, inc, s_vv, s_vh, dsig_cr, anc)
        np.savez_compressed(os.path.join(HERE, f"kernel_default_{tag}.npz"),
                            inc=inc, sigma0_vv=s_vv, sigma0_vh=s_vh, dsig_cr=dsig_cr, anc=anc,
                            **fingerprint, **out)
        print(f"kernel_default_{tag}.npz")

    # ---- G5: low-resolution LUT (resolution="low": 51 x 250 x 73, no interpolation), 32x32
    lut_co = olut.to_lut("gmf_cmod5n", resolution="low")
    lut_cr = olut.to_lut("gmf_s1_v2", resolution="low")
    luts = dict(co=lut_co.values, inc=lut_co.incidence, wspd=lut_co.wspd, phi=lut_co.phi,
                cr=lut_cr.values, wspd_cr=lut_cr.wspd)
    rng = np.random.default_rng(23)
    inc, s_vv, s_vh, dsig_cr, anc = synth_pixels(rng, (32, 32), 17.0, 65.0, np.float64)
    out = run_reference(ws, models_mod, luts, inc, s_vv, s_vh, dsig_cr, anc)
    np.savez_compressed(os.path.join(HERE, "kernel_lowres_f64.npz"),
                        inc=inc, sigma0_vv=s_vv, sigma0_vh=s_vh, dsig_cr=dsig_cr, anc=anc,
                        lut_co_sum=float(lut_co.values.sum()), lut_cr_sum=float(lut_cr.values.sum()), **out)
    print("kernel_lowres_f64.npz")


if __name__ == "__main__":
    main()
