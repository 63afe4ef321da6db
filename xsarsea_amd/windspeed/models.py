"""Model base class, registry and LUT normalisation (host side of the hot path).

Mirrors the public surface of the reference's `xsarsea/windspeed/models.py`:
`Model` (:15-301), `LutModel` (:304-347), `NcLutModel` (:350-410), `register_nc_luts` (:413-450),
`available_models` (:453-498), `get_model` (:510-538), `register_luts` (:541-568).
The LUT that feeds the device is `Model.to_lut(units="dB", **kwargs)`; unlike the reference (which
rebuilds it on every call) results are memoised per (model, kwargs).
"""
import glob
import logging
import os
from abc import abstractmethod

import numpy as np
import pandas as pd

from .lut import Lut, axis_grid, lerp_axis, xr

logger = logging.getLogger("xsarsea.windspeed.models")

_STEP_DEFAULTS = dict(inc_step_lr=1.0, wspd_step_lr=0.2, phi_step_lr=2.5, inc_step=0.1, wspd_step=0.1, phi_step=1.0)


def _interp(lut, inc, wspd, phi):
    """incidence -> wspd -> phi linear interpolation of a Lut; device kernel (xsw_lut_interp) or numpy,
    same arithmetic and bits either way (tests/test_gpu_api.py::test_lut_interp_device_equals_host)."""
    from .. import _lib, options
    mode = options.lut_interp
    if mode == "device" or (mode == "auto" and _lib.device_count_safe() > 0):
        ctx = _lib.default_context(options.device)
        return ctx.lut_interp(lut.values, lut.incidence, lut.wspd, lut.phi, inc, wspd, phi)
    vals = lerp_axis(lut.values, lut.incidence, inc, 0)
    vals = lerp_axis(vals, lut.wspd, wspd, 1)
    if lut.phi is not None:
        vals = lerp_axis(vals, lut.phi, phi, 2)
    return vals


class Model:
    """Abstract GMF/LUT model.  Registered instances are listed by `available_models()`."""

    _available_models = {}
    _name_prefix = ""
    _priority = None

    @abstractmethod
    def __init__(self, name, **kwargs):
        self.name = name
        self.pol = kwargs.pop("pol", None)
        self.units = kwargs.pop("units", None)
        self.phi_range = kwargs.pop("phi_range", None)
        self.wspd_range = kwargs.pop("wspd_range", None)
        steps = {k: kwargs.pop(k, v) for k, v in _STEP_DEFAULTS.items()}
        self.resolution = kwargs.pop("resolution", None)
        self.inc_range = kwargs.pop("inc_range", None) or [16.0, 66.0]
        self.__dict__.update(kwargs)  # free-form extras, as the reference keeps them
        self.__dict__.update(steps)
        self._lut_cache = {}
        Model._available_models[name] = self
        logger.debug("register model %s pol=%s units=%s inc=%s wspd=%s phi=%s", name, self.pol, self.units,
                     self.inc_range, self.wspd_range, self.phi_range)

    # ------------------------------------------------------------------ identity
    @property
    def short_name(self):
        prefix = type(self)._name_prefix
        if prefix and self.name.startswith(prefix):
            return self.name.replace(prefix, "", 1)
        return None

    @property
    def iscopol(self):
        """True for VV / HH models."""
        return len(set(self.pol)) == 1

    @property
    def iscrosspol(self):
        """True for VH / HV models."""
        return len(set(self.pol)) == 2

    def __repr__(self):
        return f"<{self.__class__.__name__}('{self.name}') pol={self.pol}>"

    # ------------------------------------------------------------------ LUT pipeline
    @abstractmethod
    def _raw_lut(self, **kwargs):
        """Return the model's native `Lut` (any resolution), attrs units + resolution set."""

    def _target_axes(self, resolution, kwargs):
        sfx = "" if resolution == "high" else "_lr"
        steps = [kwargs.get(f"{n}_step{sfx}", getattr(self, f"{n}_step{sfx}")) for n in ("inc", "wspd", "phi")]
        return [axis_grid(r, st) for r, st in zip((self.inc_range, self.wspd_range, self.phi_range), steps)]

    def _normalized_axes(self, have, generated_steps, has_phi, **kwargs):
        """Resolution policy of `_normalize_lut` (models.py:107-173) on axes alone: the target (inc, wspd, phi) axes when a
        LUT generated at resolution `have` (with `generated_steps`, or None) must be interpolated, None when it is used as is."""
        resolution = kwargs.get("resolution") or "high"
        do_interp = False
        if resolution == have:
            sfx = "" if resolution == "high" else "_lr"
            names = ["inc", "wspd"] + (["phi"] if self.iscopol else [])
            own = generated_steps or {n: getattr(self, f"{n}_step{sfx}") for n in names}
            do_interp = any(own[n] != kwargs.get(f"{n}_step{sfx}", own[n]) for n in names)
        if resolution == have and not do_interp:
            return None
        inc, wspd, phi = self._target_axes(resolution, kwargs)
        return inc, wspd, (phi if has_phi else None)

    def _normalize_lut(self, lut, **kwargs):
        """Bring `lut` to the requested resolution (default "high") by separable linear interpolation,
        in the LUT's own units, exactly when the reference does (models.py:107-173)."""
        lut = Lut.from_any(lut)
        resolution = kwargs.get("resolution") or "high"
        target = self._normalized_axes(lut.attrs["resolution"], lut.attrs.get("generated_steps"), lut.phi is not None, **kwargs)
        if target is None:
            return lut
        inc, wspd, phi = target
        if lut.phi is None or phi is None:
            phi = lut.phi
        vals = _interp(lut, inc, wspd, phi)
        attrs = {k: v for k, v in lut.attrs.items() if k != "generated_steps"}
        return Lut(vals, inc, wspd, phi, **{**attrs, "resolution": resolution})

    def _lut(self, units="linear", **kwargs):
        """`to_lut` on the internal container, memoised."""
        key = (units,) + tuple(sorted(kwargs.items()))
        hit = self._lut_cache.get(key)
        if hit is not None:
            return hit
        lut = self._normalize_lut(self._raw_lut(**kwargs), **kwargs)
        have = lut.attrs["units"]
        if units is not None and units != have:
            if units == "dB":
                lut = lut.with_values(10 * np.log10(lut.values + 1e-15), units="dB")  # models.py:210-216
            elif units == "linear":
                lut = lut.with_values(10.0 ** (lut.values / 10.0), units="linear")  # models.py:217-222
            else:
                raise ValueError(f"Unit not known: {units}. Known are 'dB' or 'linear' ")
        elif units not in (None, "dB", "linear"):
            raise ValueError(f"Unit not known: {units}. Known are 'dB' or 'linear' ")
        lut.attrs["model"] = self.name
        lut.attrs["pol"] = self.pol
        self._lut_cache[key] = lut
        return lut

    def to_lut(self, units="linear", **kwargs):
        """Model LUT in `units` ('linear' / 'dB' / None).  kwargs: `resolution` ("low"/"high"/None) and
        the `*_step` / `*_step_lr` overrides.  Returns an xarray.DataArray when xarray is installed
        (the reference's type), else the numpy-backed `Lut`."""
        lut = self._lut(units=units, **kwargs)
        return lut.to_xarray() if xr is not None else lut

    def to_netcdf(self, file):
        """Save the model as an xsarsea-format netCDF LUT (models.py:232-262): through xarray when it is installed, else as
        classic netCDF-3 through scipy (`nc_io.write_lut`: same variables and global attributes)."""
        resolution = "low" if self.iscopol else "high"
        lut = self._lut(resolution=resolution, units="dB")
        if xr is None:
            from . import nc_io
            step = lambda ax: float(np.round(np.unique(np.diff(ax)), decimals=2)[0])
            attrs = dict(units="dB", resolution=resolution, model=self.short_name, pol=self.pol, inc_range=self.inc_range,
                         wspd_range=self.wspd_range, wspd_step=step(lut.wspd), inc_step=step(lut.incidence))
            if lut.phi is not None:
                attrs.update(phi_range=self.phi_range, phi_step=step(lut.phi))
            return nc_io.write_lut(file, lut, attrs)
        ds = lut.to_xarray().to_dataset(promote_attrs=True)
        ds.sigma0_model.attrs.clear()
        ds.attrs.update(pol=self.pol, inc_range=self.inc_range, wspd_range=self.wspd_range, resolution=resolution,
                        model=self.short_name)
        ds.attrs["wspd_step"] = np.round(np.unique(np.diff(lut.wspd)), decimals=2)[0]
        ds.attrs["inc_step"] = np.round(np.unique(np.diff(lut.incidence)), decimals=2)[0]
        if lut.phi is not None:
            ds.attrs["phi_range"] = self.phi_range
            ds.attrs["phi_step"] = np.round(np.unique(np.diff(lut.phi)), decimals=2)[0]
        ds.to_netcdf(file)

    @abstractmethod
    def __call__(self, inc, wspd, phi=None, broadcast=False):
        raise NotImplementedError(self.__class__)


class LutModel(Model):
    """Model defined by a stored table; evaluation = multilinear interpolation of the table."""

    _name_prefix = "nc_lut_"
    _priority = None

    def __call__(self, inc, wspd, phi=None, units=None, **kwargs):
        vals = [v for v in (inc, wspd, phi) if v is not None]
        all_scalar = all(np.isscalar(v) for v in vals)
        all_1d = not all_scalar and all(getattr(v, "ndim", None) == 1 for v in vals)
        if not (all_scalar or all_1d):
            raise NotImplementedError("Only scalar or 1D array are implemented for LutModel")
        kwargs.pop("broadcast", None)
        lut = self._lut(units=units, **kwargs)
        # plain `lut.interp(...)` in the reference (models.py:330-346): points outside the table are NaN, not an error
        out = lerp_axis(lut.values, lut.incidence, np.atleast_1d(np.asarray(inc, dtype=np.float64)), 0, bounds_error=False)
        out = lerp_axis(out, lut.wspd, np.atleast_1d(np.asarray(wspd, dtype=np.float64)), 1, bounds_error=False)
        if lut.phi is not None:
            out = lerp_axis(out, lut.phi, np.atleast_1d(np.asarray(phi, dtype=np.float64)), 2, bounds_error=False)
        if all_scalar:
            return out.item()
        if xr is not None:
            dims = lut.dims
            coords = dict(zip(dims, (inc, wspd, phi)))
            da = xr.DataArray(out, dims=dims, coords={d: np.asarray(coords[d]) for d in dims}, name="sigma0_gmf")
            da.attrs.update(model=self.name, units=self.units)
            return da
        return out


class ArrayLutModel(LutModel):
    """LUT model over an in-memory table (any source: CMOD7 binary, sarwing npy, synthetic)."""

    _name_prefix = "gmf_"
    _priority = 1

    def __init__(self, name, lut, **kwargs):
        lut = Lut.from_any(lut)
        kwargs.setdefault("units", lut.attrs["units"])
        kwargs.setdefault("resolution", lut.attrs["resolution"])
        kwargs.setdefault("inc_range", [float(lut.incidence[0]), float(lut.incidence[-1])])
        kwargs.setdefault("wspd_range", [float(lut.wspd[0]), float(lut.wspd[-1])])
        if lut.phi is not None:
            kwargs.setdefault("phi_range", [float(lut.phi[0]), float(lut.phi[-1])])
        sfx = "_lr" if lut.attrs["resolution"] == "low" else ""
        for n, ax in (("inc", lut.incidence), ("wspd", lut.wspd), ("phi", lut.phi)):
            if ax is not None and len(ax) > 1:
                kwargs.setdefault(f"{n}_step{sfx}", float((ax[-1] - ax[0]) / (len(ax) - 1)))
        super().__init__(name, **kwargs)
        self._table = lut

    def _raw_lut(self, **kwargs):
        return self._table


class NcLutModel(LutModel):
    """LUT stored in the xsarsea netCDF format (variable `sigma0_model`, global attrs units / pol /
    model / resolution / *_range / *_step; models.py:361-410).  Read through xarray when it is installed, else through
    `nc_io`: classic netCDF-3 files with scipy, netCDF-4 / HDF5 files with the package's own minimal HDF5 reader."""

    _priority = 10

    @property
    def short_name(self):
        return self._short_name

    def __init__(self, path, **kwargs):
        name = os.path.splitext(os.path.basename(path))[0]
        if xr is None:  # nc_io: classic netCDF-3 through scipy, netCDF-4 / HDF5 through hdf5_min
            from . import nc_io
            file_attrs = nc_io.read_attrs(path)
        else:
            with xr.open_dataset(path) as nc:
                file_attrs = dict(nc.attrs)
        for attr in ("units", "pol", "model", "resolution", "inc_range", "wspd_range", "phi_range", "inc_step",
                     "wspd_step", "phi_step"):
            if attr in file_attrs:
                v = file_attrs[attr]
                kwargs[attr] = [float(x) for x in v] if isinstance(v, np.ndarray) else v
        self._short_name = kwargs.pop("model")
        if kwargs["resolution"] == "low":
            kwargs["inc_step_lr"] = kwargs.pop("inc_step")
            kwargs["wspd_step_lr"] = kwargs.pop("wspd_step")
            if kwargs.get("phi_step") is not None:
                kwargs["phi_step_lr"] = kwargs.pop("phi_step")
            else:
                kwargs.pop("phi_step", None)
        super().__init__(name, **kwargs)
        self.path = path

    def _raw_lut(self, **kwargs):
        if not os.path.isfile(self.path):
            raise FileNotFoundError(self.path)
        if xr is None:
            from . import nc_io
            return nc_io.read_lut(self.path)
        ds = xr.open_dataset(self.path)
        da = ds.sigma0_model
        da.attrs["units"] = ds.attrs["units"]
        da.attrs["model"] = ds.attrs["model"]
        da.attrs["resolution"] = ds.attrs["resolution"]
        return Lut.from_any(da)


def register_nc_luts(topdir, gmf_names=None):
    """Register every `nc_lut_*.nc` under `topdir` (optionally filtered by name)."""
    for path in glob.glob(os.path.join(topdir, f"{NcLutModel._name_prefix}*.nc")):
        path = os.path.abspath(path)
        name = os.path.basename(path).replace(".nc", "")
        if gmf_names is None or name in gmf_names:
            NcLutModel(path)


def available_models(pol=None):
    """DataFrame (index = model name; columns alias, pol, model) of the registered models.  The alias is
    the short name, owned by the highest-priority (lowest number) model carrying it."""
    rows = [dict(name=n, alias=m.short_name, priority=m._priority, pol=m.pol, model=m)
            for n, m in Model._available_models.items()]
    df = pd.DataFrame(rows, columns=["name", "alias", "priority", "pol", "model"]).set_index("name")
    df.index.name = None
    if len(df):
        order = df.sort_values("priority", ascending=True, kind="stable")
        owners = order.drop_duplicates("alias").index
        aliased = order.loc[owners]
        rest = df.drop(owners).copy()
        rest["alias"] = None
        df = pd.concat([aliased, rest])
    df = df.drop(columns="priority")
    if pol is not None:
        df = df[df.pol == pol]
    return df


def get_model(name):
    """Model by name or alias; a Model instance is passed through.  KeyError if unknown."""
    if isinstance(name, Model):
        return name
    models = Model._available_models
    if name in models:
        return models[name]
    df = available_models()
    hit = df[df.alias == name]
    if len(hit) != 1:
        raise KeyError(f"model {name} not found")
    return hit.model.iloc[0]


def register_luts(topdir=None, topdir_cmod7=None):
    """Activate the analytic GMFs, then the netCDF LUTs of `topdir` and CMOD7 from `topdir_cmod7`."""
    from . import cmod7, gmfs

    gmfs.GmfModel.activate_gmfs_impl()
    if topdir is not None:
        register_nc_luts(topdir)
    if topdir_cmod7 is not None:
        cmod7.register_cmod7(topdir_cmod7)
