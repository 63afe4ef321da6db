"""Dense model LUT container used on the host side of the hot path.

The reference passes LUTs around as `xarray.DataArray` with dims (incidence, wspd[, phi]) and attrs
`units` / `resolution` (windspeed/models.py:82-105).  The device only needs the dense float64 block
and its axes, so the product keeps a small numpy-backed container and converts to/from xarray at
the API edge when xarray is installed.
"""
import numpy as np

try:  # optional: only needed to hand LUTs to users as DataArray
    import xarray as xr
except ImportError:  # pragma: no cover - xarray is absent from the build image
    xr = None

DIMS2 = ("incidence", "wspd")
DIMS3 = ("incidence", "wspd", "phi")
ALLOWED_UNITS = ("linear", "dB")


class Lut:
    """values[incidence, wspd(, phi)] float64 + axes + attrs (units, resolution, model, pol)."""

    def __init__(self, values, incidence, wspd, phi=None, units="linear", resolution=None, **attrs):
        self.values = np.asarray(values, dtype=np.float64)
        self.incidence = np.asarray(incidence, dtype=np.float64)
        self.wspd = np.asarray(wspd, dtype=np.float64)
        self.phi = None if phi is None else np.asarray(phi, dtype=np.float64)
        self.attrs = dict(units=units, resolution=resolution, **attrs)
        self.name = "sigma0_model"
        expect = (len(self.incidence), len(self.wspd)) + (() if phi is None else (len(self.phi),))
        if self.values.shape != expect:
            raise IndexError(f"Bad LUT shape {self.values.shape}, axes say {expect}")

    # -- DataArray-like surface used by callers of to_lut()
    @property
    def dims(self):
        return DIMS2 if self.phi is None else DIMS3

    @property
    def ndim(self):
        return self.values.ndim

    @property
    def shape(self):
        return self.values.shape

    def __array__(self, dtype=None, copy=None):
        return self.values if dtype is None else self.values.astype(dtype)

    def transpose(self, *dims):
        order = [self.dims.index(d) for d in dims]
        return np.transpose(self.values, order)

    def with_values(self, values, **attr_updates):
        out = Lut(values, self.incidence, self.wspd, self.phi, **{**self.attrs, **attr_updates})
        return out

    def to_xarray(self):
        if xr is None:
            raise ImportError("xarray is not installed")
        coords = {"incidence": self.incidence, "wspd": self.wspd}
        if self.phi is not None:
            coords["phi"] = self.phi
        da = xr.DataArray(self.values, dims=self.dims, coords=coords, name=self.name)
        da.attrs.update({k: v for k, v in self.attrs.items() if v is not None})
        return da

    @classmethod
    def from_any(cls, lut):
        """Accept a Lut or an xarray.DataArray laid out like the reference's LUTs and validate it the
        way Model._normalize_lut does (models.py:84-105): KeyError / ValueError / IndexError."""
        if isinstance(lut, cls):
            out = lut
        else:
            attrs = dict(getattr(lut, "attrs", {}))
            dims = tuple(getattr(lut, "dims", ()))
            if dims not in (DIMS2, DIMS3):
                raise IndexError(f"Bad dims '{dims}'. Should be '{DIMS2}' or '{DIMS3}'")
            phi = np.asarray(lut["phi"]) if "phi" in dims else None
            units = attrs.pop("units", None)
            resolution = attrs.pop("resolution", None)
            out = cls(np.asarray(lut), np.asarray(lut["incidence"]), np.asarray(lut["wspd"]), phi,
                      units=units, resolution=resolution, **attrs)
        if out.attrs.get("units") is None:
            raise KeyError("lut has no lut.attrs['units']")
        if out.attrs["units"] not in ALLOWED_UNITS:
            raise ValueError(f"Unknown lut units '{out.attrs['units']}'. Allowed are '{list(ALLOWED_UNITS)}'")
        assert out.attrs.get("resolution") is not None, "lut has no attrs['resolution']"
        return out


def axis_grid(rng, step):
    """Axis of a generated LUT: linspace with round((hi-lo)/step)+1 points (models.py:154-160)."""
    if rng is None:
        return None
    return np.linspace(rng[0], rng[1], num=int(np.round((rng[1] - rng[0]) / step) + 1))


def lerp_axis(values, x_old, x_new, axis, bounds_error=True):
    """Linear interpolation of `values` along `axis` from grid x_old to x_new.

    Same arithmetic as scipy.interpolate.interp1d(kind="linear"), which is what the reference's
    `lut.interp(...)` runs once per dimension: slope = (y_hi - y_lo)/(x_hi - x_lo),
    y = slope*(x_new - x_lo) + y_lo.  bounds_error=True is `_normalize_lut`'s call (models.py:167: ValueError
    outside the table); bounds_error=False is `LutModel.__call__`'s plain `lut.interp(...)` (models.py:330-346: NaN
    outside the table, NaN abscissae give NaN).
    """
    x_old = np.asarray(x_old, dtype=np.float64)
    x_new = np.asarray(x_new, dtype=np.float64)
    outside = ~((x_new >= x_old[0]) & (x_new <= x_old[-1]))
    if bounds_error and x_new.size and outside.any():
        raise ValueError("A value in x_new is outside the interpolation range.")
    hi = np.clip(np.searchsorted(x_old, x_new), 1, len(x_old) - 1)
    lo = hi - 1
    v = np.moveaxis(values, axis, 0)
    bshape = (-1,) + (1,) * (v.ndim - 1)
    y_lo, y_hi = v[lo], v[hi]
    slope = (y_hi - y_lo) / (x_old[hi] - x_old[lo]).reshape(bshape)
    out = slope * (x_new - x_old[lo]).reshape(bshape) + y_lo
    if not bounds_error and outside.any():
        out[outside] = np.nan
    return np.moveaxis(out, 0, axis)
