"""GmfModel: models defined by an analytic function sigma0 = f(inc, wspd[, phi]).

Mirrors the plugin surface of the reference's `xsarsea/windspeed/gmfs.py`: the `GmfModel.register`
decorator (:23-105), deferred activation (:112-125), phi-range probing at construction (:127-168),
evaluation on scalars / 1-D grids / broadcast arrays (`__call__`, :266-348) and the raw LUT on
`linspace` grids (`_raw_lut`, :350-395).

Where the reference JIT-compiles the scalar function with numba (njit / vectorize / guvectorize,
:202-236), this build evaluates the function on whole numpy arrays: the built-in GMFs
(`gmfs_impl.py`) are written array-wise; a user function that only accepts scalars is wrapped
with numpy.vectorize.
"""
import logging

import numpy as np

from .lut import Lut, axis_grid, xr
from .models import Model

logger = logging.getLogger("xsarsea.windspeed")


def _array_eval(func, inc, wspd, phi):
    """Call `func` on broadcastable arrays; fall back to element-wise calls for scalar-only functions."""
    try:
        with np.errstate(all="ignore"):
            out = func(inc, wspd, phi)
        out = np.asarray(out, dtype=np.float64)
        shape = np.broadcast_shapes(*(np.shape(v) for v in (inc, wspd, phi) if v is not None))
        if out.shape == shape or out.size == int(np.prod(shape)):
            return np.broadcast_to(out, shape) if out.shape != shape else out
        raise ValueError("shape")
    except (ValueError, TypeError):
        vec = np.vectorize(lambda a, b, c: func(float(a), float(b), None if np.isnan(c) else float(c)), otypes=[np.float64])
        with np.errstate(all="ignore"):
            return vec(inc, wspd, np.nan if phi is None else phi)


class GmfModel(Model):
    """Model backed by an analytic GMF.  See `GmfModel.register`."""

    _name_prefix = "gmf_"
    _priority = 3
    _registry = {}
    _deferred_registrations = []

    @classmethod
    def register(cls, name=None, pol=None, units="linear", defer=True, **kwargs):
        """Decorator registering `func(inc, wspd[, phi])` as model `name` (default: the function's name,
        which must start with 'gmf_').  `wspd_range` defaults to [0.2, 50] (co-pol) or [3, 80]
        (cross-pol).  With defer=True the model only appears after `activate_gmfs_impl()`."""

        def inner(func):
            gmf_name = name or func.__name__
            if not gmf_name.startswith(cls._name_prefix):
                raise ValueError(f"gmf function must start with '{cls._name_prefix}'. Got {gmf_name}")
            wspd_range = kwargs.pop("wspd_range", None)
            if wspd_range is None:
                wspd_range = [0.2, 50.0] if len(set(pol)) == 1 else [3.0, 80.0]
            if defer:
                cls._deferred_registrations.append((func, gmf_name, wspd_range, pol, units, kwargs))
            else:
                cls._register_function(func, gmf_name, wspd_range, pol, units, **kwargs)
            return func

        return inner

    @classmethod
    def _register_function(cls, func, name, wspd_range, pol, units, **kwargs):
        cls._registry[name] = cls(name, func, wspd_range, pol, units, **kwargs)

    @classmethod
    def activate_gmfs_impl(cls, gmfs_names=None, **kwargs):
        """Instantiate deferred registrations (all, or only `gmfs_names`)."""
        for func, name, wspd_range, pol, units, reg_kwargs in cls._deferred_registrations:
            if gmfs_names is None or name in gmfs_names:
                cls._register_function(func, name, wspd_range, pol, units, **{**reg_kwargs, **kwargs})

    def __init__(self, name, gmf_pyfunc_scalar, wspd_range=(0.2, 50.0), pol=None, units=None, **kwargs):
        # scalar probe: a GMF that only takes arrays raises TypeError here, as in the reference
        probe = [gmf_pyfunc_scalar(35.0, 0.2, 90.0)]
        try:
            gmf_pyfunc_scalar(35.0, 0.2, None)
            phi_range = None
        except TypeError:
            # direction-dependent: symmetric in +-phi  ->  LUT over [0, 180], else [0, 360]
            probe = [np.abs(gmf_pyfunc_scalar(35.0, 0.2, p) - gmf_pyfunc_scalar(35.0, 0.2, -p)) for p in (0, 90, 180, 270)]
            phi_range = [0.0, 180.0] if min(probe) < 1e-15 else [0.0, 360.0]
        if (units == "dB" and min(probe) > 0) or (units == "linear" and min(probe) < 0):
            logger.info(f"Possible bad units '{units}'  for gmf {name}")
        super().__init__(name, units=units, pol=pol, wspd_range=list(wspd_range), phi_range=phi_range, **kwargs)
        self._gmf_pyfunc_scalar = gmf_pyfunc_scalar

    # ------------------------------------------------------------------ evaluation
    def __call__(self, inc, wspd, phi=None, broadcast=False, numba=True):
        """sigma0 (model units).  All scalars -> scalar; all 1-D -> grid (incidence, wspd[, phi]);
        otherwise (or broadcast=True) inputs are broadcast and the result has their common shape."""
        args = [v for v in (inc, wspd, phi) if v is not None]
        all_scalar = all(np.isscalar(v) for v in args)
        all_1d = all(getattr(v, "ndim", None) == 1 for v in args)
        if any(getattr(v, "ndim", 0) > 1 for v in args):
            broadcast = True
        template = None
        if broadcast:
            for v in (inc, wspd, phi):
                if xr is not None and isinstance(v, xr.DataArray):
                    template = v
                    break
            arrs = np.broadcast_arrays(*[np.asarray(v, dtype=np.float64) for v in args])
            out = self._broadcast_eval(arrs[0], arrs[1], arrs[2] if phi is not None else None)
            if template is not None and template.shape == out.shape:
                res = template.copy().astype(np.float64)
                res.attrs.clear()
                res.data = out
                res.attrs["units"] = self.units
                return res
            return out
        if all_scalar:
            return float(_array_eval(self._gmf_pyfunc_scalar, np.float64(inc), np.float64(wspd),
                                     None if phi is None else np.float64(phi)))
        if all_1d:
            grid = self._grid(np.asarray(inc, dtype=np.float64), np.asarray(wspd, dtype=np.float64),
                              None if phi is None else np.asarray(phi, dtype=np.float64))
            if xr is None:
                return grid
            names = ["incidence", "wspd"] + (["phi"] if phi is not None else [])
            coords = {n: np.asarray(v) for n, v in zip(names, args)}
            da = xr.DataArray(grid, dims=names, coords=coords)
            da.attrs["units"] = self.units
            return da
        raise ValueError("Non 1d shape must all have the same shape")

    def _broadcast_eval(self, inc, wspd, phi):
        """Elementwise evaluation on same-shape arrays: device kernel (xsw_gmf_eval) for large inputs of a
        built-in model, numpy otherwise."""
        from .. import _lib, options
        gid = _lib.GMF_IDS.get(self.name) if getattr(self, "_builtin", False) else None
        mode = options.gmf_on_device
        if gid is not None and (mode == "device" or (mode == "auto" and inc.size >= options.gmf_device_min_size
                                                     and _lib.device_count_safe() > 0)):
            return _lib.default_context(options.device).gmf_eval(gid, inc, wspd, phi)
        return _array_eval(self._gmf_pyfunc_scalar, inc, wspd, phi)

    def _grid(self, inc, wspd, phi):
        """Dense (incidence, wspd[, phi]) evaluation: the fill of gmfs.py:215-232."""
        if phi is None:
            return np.ascontiguousarray(_array_eval(self._gmf_pyfunc_scalar, inc[:, None], wspd[None, :], None))
        return np.ascontiguousarray(
            _array_eval(self._gmf_pyfunc_scalar, inc[:, None, None], wspd[None, :, None], phi[None, None, :]))

    def _raw_axes(self, **kwargs):
        """Grid the raw LUT is generated on (gmfs.py:350-384): (inc, wspd, phi, resolution, generated steps)."""
        resolution = kwargs.get("resolution", "low")  # generated at low resolution by default (:353)
        if resolution not in ("low", "high", None):
            raise ValueError('kwargs resolution must be "low" or "high" or None, or not provided')
        if resolution is None:
            resolution = "low" if self.iscopol else "high"
        sfx = "_lr" if resolution == "low" else ""
        # The reference overwrites self.<axis>_step<sfx> with the steps it generated with (gmfs.py:370-379) so
        # that _normalize_lut sees "already at the requested steps"; that also leaks one call's kwargs into
        # every later call.  Here the generated steps travel with the LUT instead (same decision, no leak).
        steps = {n: kwargs.get(f"{n}_step{sfx}", getattr(self, f"{n}_step{sfx}")) for n in ("inc", "wspd", "phi")}
        inc, wspd, phi = (axis_grid(r, steps[n]) for r, n in
                          zip((self.inc_range, self.wspd_range, self.phi_range), ("inc", "wspd", "phi")))
        return inc, wspd, phi, resolution, steps

    def _raw_lut(self, **kwargs):
        inc, wspd, phi, resolution, steps = self._raw_axes(**kwargs)
        return Lut(self._grid(inc, wspd, phi), inc, wspd, phi, units=self.units, resolution=resolution,
                   generated_steps=steps)

    def device_lut_plan(self, **kwargs):
        """(gmf_id, raw axes, target axes) for `xsw_lut_build`, or None when this model / these kwargs cannot be built on the
        device (not a built-in GMF, not in linear units).  Same grid policy as `_raw_lut` + `Model._normalize_lut`."""
        from .. import _lib
        gid = _lib.GMF_IDS.get(self.name) if getattr(self, "_builtin", False) else None
        if gid is None or self.units != "linear":
            return None
        inc, wspd, phi, have, steps = self._raw_axes(**kwargs)
        target = self._normalized_axes(have, steps, phi is not None, **kwargs)
        if target is None:
            target = (inc, wspd, phi)
        return gid, (inc, wspd, phi), target
