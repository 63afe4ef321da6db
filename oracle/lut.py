"""Oracle restatement of the reference's LUT preparation (test infrastructure).

Restates, for GMF-backed models:
  * grid construction ......... models.py:154-160 / gmfs.py:387-392
        np.linspace(r0, r1, num=int(np.round((r1 - r0) / step) + 1))
  * raw LUT ................... gmfs.py:350-395  (default resolution "low": models.py:42-44 steps
                                inc 1 deg, wspd 0.2 m/s, phi 2.5 deg; evaluated by the guvectorize
                                triple loop gmfs.py:215-232 -> shape (incidence, wspd, phi))
  * resolution policy ......... models.py:107-152 (_normalize_lut: default target "high":
                                inc 0.1, wspd 0.1, phi 1.0 from models.py:46-48)
  * interpolation ............. models.py:167  lut.interp(incidence=, wspd=, phi=,
                                kwargs=dict(bounds_error=True)) == xarray's orthogonal decomposition
                                into sequential scipy.interpolate.interp1d(kind="linear") calls, in
                                the order incidence -> wspd -> phi, performed in LINEAR units
  * dB conversion ............. models.py:210-216   10*log10(lut + 1e-15)

xarray is not installed in the build image, so the bit-equality of the interpolation with the
reference's `DataArray.interp` is UNPINNED (formula-pinned); everything downstream (the search
kernel) is tested on explicit LUT arrays and does not depend on it.
"""
import numpy as np
from scipy.interpolate import interp1d

from . import gmf as _gmf

INC_RANGE = [16.0, 66.0]  # models.py:38-40
STEPS_LOW = dict(inc=1.0, wspd=0.2, phi=2.5)  # models.py:42-44
STEPS_HIGH = dict(inc=0.1, wspd=0.1, phi=1.0)  # models.py:46-48


def grid(r, step):
    """models.py:154-160 / gmfs.py:387-392."""
    if r is None:
        return None
    return np.linspace(r[0], r[1], num=int(np.round((r[1] - r[0]) / step) + 1))


class Lut:
    """Plain container standing in for the reference's xarray LUT (dims incidence, wspd[, phi])."""

    def __init__(self, values, incidence, wspd, phi, units, resolution, name, pol):
        self.values = values
        self.incidence = incidence
        self.wspd = wspd
        self.phi = phi
        self.units = units
        self.resolution = resolution
        self.name = name
        self.pol = pol

    @property
    def iscopol(self):  # models.py:176-179
        return len(set(self.pol)) == 1


def raw_lut(name, resolution="low", inc_range=None, gmf_func=None, pol=None, wspd_range=None,
            phi_range=None, **steps):
    """gmfs.py:350-395 (GmfModel._raw_lut), linear units."""
    if gmf_func is None:
        gmf_func, pol, wspd_range, phi_range = _gmf.GMFS[name]
    inc_range = inc_range or INC_RANGE
    iscopol = len(set(pol)) == 1
    if resolution is None:  # gmfs.py:357-362
        resolution = "low" if iscopol else "high"
    base = STEPS_LOW if resolution == "low" else STEPS_HIGH
    sfx = "_lr" if resolution == "low" else ""
    inc_step = steps.get("inc_step" + sfx, base["inc"])
    wspd_step = steps.get("wspd_step" + sfx, base["wspd"])
    phi_step = steps.get("phi_step" + sfx, base["phi"])
    inc = grid(inc_range, inc_step)
    wspd = grid(wspd_range, wspd_step)
    phi = grid(phi_range, phi_step)
    if phi is not None:
        vals = gmf_func(inc[:, None, None], wspd[None, :, None], phi[None, None, :])
    else:
        vals = gmf_func(inc[:, None], wspd[None, :], None)
    vals = np.ascontiguousarray(np.broadcast_to(vals, vals.shape), dtype=np.float64)
    return Lut(vals, inc, wspd, phi, "linear", resolution, name, pol)


def interp_axis(values, x_old, x_new, axis):
    """One leg of xarray's orthogonal linear interpolation: scipy interp1d, bounds_error=True."""
    f = interp1d(x_old, values, kind="linear", axis=axis, bounds_error=True, assume_sorted=True)
    return f(x_new)


def normalize_lut(lut, inc_range=None, wspd_range=None, phi_range=None, resolution="high", **steps):
    """models.py:82-174 for the cases the hot path exercises (low->high, or no interpolation)."""
    if resolution is None:
        resolution = "high"
    if resolution == lut.resolution:
        return lut  # same resolution, same steps: no interpolation (models.py:118-132, :170-173)
    base = STEPS_HIGH if resolution == "high" else STEPS_LOW
    sfx = "" if resolution == "high" else "_lr"
    inc = grid(inc_range or INC_RANGE, steps.get("inc_step" + sfx, base["inc"]))
    wspd = grid(wspd_range or [lut.wspd[0], lut.wspd[-1]], steps.get("wspd_step" + sfx, base["wspd"]))
    vals = interp_axis(lut.values, lut.incidence, inc, 0)
    vals = interp_axis(vals, lut.wspd, wspd, 1)
    phi = None
    if lut.phi is not None:
        phi = grid(phi_range or [lut.phi[0], lut.phi[-1]], steps.get("phi_step" + sfx, base["phi"]))
        vals = interp_axis(vals, lut.phi, phi, 2)
    return Lut(np.ascontiguousarray(vals), inc, wspd, phi, lut.units, resolution, lut.name, lut.pol)


def to_lut(name, units="dB", **kwargs):
    """models.py:186-230 (Model.to_lut) for a registered analytic GMF `name`.

    kwargs are the reference's `**kwargs` of invert_from_model (windspeed.py:144): `resolution`
    ("low"/"high"/None) and the step overrides.
    """
    _, pol, wspd_range, phi_range = _gmf.GMFS[name]
    raw_kw = dict(kwargs)
    lut = raw_lut(name, resolution=raw_kw.pop("resolution", "low"), **raw_kw)
    norm_kw = dict(kwargs)
    lut = normalize_lut(lut, INC_RANGE, wspd_range, phi_range,
                        resolution=norm_kw.pop("resolution", "high"), **norm_kw)
    if units == "dB" and lut.units == "linear":
        lut = Lut(10 * np.log10(lut.values + 1e-15), lut.incidence, lut.wspd, lut.phi, "dB",
                  lut.resolution, lut.name, lut.pol)
    elif units == "linear" and lut.units == "dB":
        lut = Lut(10.0 ** (lut.values / 10.0), lut.incidence, lut.wspd, lut.phi, "linear",
                  lut.resolution, lut.name, lut.pol)
    return lut
