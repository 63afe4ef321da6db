"""`invert_from_model`: drop-in for `xsarsea.windspeed.invert_from_model`
(reference: src/xsarsea/windspeed/windspeed.py:17-439), running the per-pixel search on MI355X.

What stays on the host, line for line in behaviour: model lookup (:78-83), mono / cross / dual
routing with its pol check, warnings and assertion (:88-120), scalar `dsig_cr` broadcasting
(:122-123), container handling (xarray in -> xarray out with name/attrs :338-343, :395-438; numpy
in -> numpy out; dask in -> lazy dask out :350-364) and the return conventions (:415-439).
What moves to the device: the sigma0 -> dB conversion (:126-130, see `options.db_on_device`) and all
of `_invert_from_model_numpy` (:132-331) through `xsw_invert` (include/xsw.h).

Beyond the reference's containers: rasters already resident in HBM (torch CUDA tensors, `__cuda_array_interface__`
objects) are inverted in place and torch tensors come back (`_engine.invert_device`); `options.devices` spreads the row
tiles of a host raster over several GPUs inside the one call (the counterpart of the reference's numba thread pool).
"""
import logging
import time
import warnings

import numpy as np

from .. import _device, options
from . import _engine
from .lut import xr
from .models import get_model

logger = logging.getLogger("xsarsea.windspeed")

try:
    import dask.array as da
except ImportError:  # pragma: no cover - dask is absent from the build image
    da = None


def _valid(v):
    """np.any(~np.isnan(v)) (windspeed.py:107, :112); numpy rasters are scanned block-wise with early exit."""
    if isinstance(v, np.ndarray):
        return _engine.any_valid(v)
    if _device.is_device_array(v):  # one reduction on the device (synchronises: the reference's assertion needs the answer)
        import torch
        return bool(torch.isnan(_device.as_tensor(v, _device.device_of(v))).logical_not().any().item())
    return bool(np.any(~np.isnan(v)))


def _is_xr(v):
    return xr is not None and isinstance(v, xr.DataArray)


def _is_dask(v):
    data = getattr(v, "data", v)
    return da is not None and isinstance(data, da.Array)


class CodedWinds:
    """The answer of an inversion of numpy rasters as 4-byte grid codes (include/xsw.h: out_code_*) + what turns them into the
    return value of `invert_from_model`: `finish()` expands them on the host (bit-identical to the direct call) and applies the
    return conventions of windspeed.py:415-439.  What `multi_gpu.invert_from_model_tiled` sends between ranks."""

    def __init__(self, mode, lut_co, lut_cr, codes_co, codes_cr, launch=None, on_device=False):
        self.mode, self.lut_co, self.lut_cr, self.codes_co, self.codes_cr = mode, lut_co, lut_cr, codes_co, codes_cr
        # the gathered multi-GPU call (`_engine.invert_coded`): the codes stay in device memory, `launch()` queues the chunked
        # inversion + gather + expansion, and the expanded winds come back through `finish_winds`
        self.launch, self.on_device = launch, on_device

    def finish(self, codes_co=None, codes_cr=None):
        cc = self.codes_co if codes_co is None else codes_co
        cr = self.codes_cr if codes_cr is None else codes_cr
        ws_co, ws_cr = _engine.expand_codes(self.lut_co, self.lut_cr, cc, cr)
        if self.mode == "mono_co":
            return ws_co
        if self.mode == "mono_cr":
            return _engine.abs_blocks(ws_cr)
        return ws_co, _engine.dual_select(ws_co, ws_cr)

    def finish_winds(self, ws_co, ws_cr):
        """The return conventions of windspeed.py:415-439 on expanded winds: numpy arrays (complex128; the dual-pol select with
        numpy's own `abs`, as the direct call does) or torch tensors (device rasters: the select was fused into the kernel)."""
        if self.mode == "mono_co":
            return ws_co
        if self.on_device:
            return ws_cr.abs() if self.mode == "mono_cr" else (ws_co, ws_cr)
        if self.mode == "mono_cr":
            return _engine.abs_blocks(ws_cr)
        return ws_co, _engine.dual_select(ws_co, ws_cr)


def invert_from_model(inc, sigma0, sigma0_dual=None, /, ancillary_wind=None, dsig_co=0.1, dsig_cr=0.1, model=None,
                      **kwargs):
    """
    Invert sigma0 to wind from a model (GMF or LUT).

    Parameters
    ----------
    inc : xarray.DataArray | numpy.ndarray
        incidence angle (deg)
    sigma0 : same type
        linear sigma0 to invert
    sigma0_dual : same type, optional
        cross-pol sigma0 for the dual-pol inversion
    ancillary_wind : complex array, optional
        a-priori wind in **antenna convention** (real = sample axis, imag = line axis)
    dsig_co : float
        `Jsig_co = ((sigma0_gmf - sigma0) / dsig_co) ** 2`
    dsig_cr : float or array
        `Jsig_cr = ((sigma0_gmf - sigma0) / dsig_cr) ** 2`
    model : str | Model | (co, cross) tuple
    **kwargs : forwarded to `Model.to_lut` (`resolution`, `inc_step`, ...)

    Returns
    -------
    co-pol model: complex wind (antenna convention);  cross-pol model: wind speed (float);
    dual-pol: `(wind_co, wind_dual)`.  The container type follows the inputs.
    """
    t0 = time.time()
    # private: set by multi_gpu.invert_from_model_tiled, whose per-rank call sees one row tile of the raster -- the ancillary-wind
    # precondition is a whole-raster property (windspeed.py:107, :112) and arrives as the all-reduced answer
    tile_any_valid = kwargs.pop("_xsw_tile", None)
    want_codes = kwargs.pop("_xsw_codes", False)  # private (multi_gpu): numpy rasters -> `CodedWinds` instead of the winds
    models = model if isinstance(model, tuple) else (model, None)
    models = tuple(get_model(m) if m is not None else None for m in models)
    no_ancillary = ancillary_wind is None  # the reference substitutes an all-NaN array (sigma0 * nan, :71-86)

    if sigma0_dual is None:
        try:
            pol = sigma0.pol.values.item()
        except AttributeError:
            pol = None
        model_pol = models[0].pol
        if pol is None:
            warnings.warn(f"Unable to check sigma0 pol. Assuming  {model_pol}")
        elif pol not in model_pol:
            raise ValueError(f"sigma0 pol is {pol}, and model {models[0].name} can only handle {model_pol}")
        if models[0].iscopol:
            sigma0_co, sigma0_cr = sigma0, None
            assert not no_ancillary and (_valid(ancillary_wind) if tile_any_valid is None else tile_any_valid), \
                "co-pol inversion needs a valid ancillary wind"
        elif models[0].iscrosspol:
            sigma0_co, sigma0_cr = None, sigma0
            if not no_ancillary and (_valid(ancillary_wind) if tile_any_valid is None else tile_any_valid):
                warnings.warn("crosspol inversion is best without ancillary wind, but using it as requested.")
            models = (None, models[0])
    else:
        sigma0_co, sigma0_cr = sigma0, sigma0_dual

    lut_co = _engine.lut_source(models[0], kwargs) if models[0] is not None else None
    lut_cr = _engine.lut_source(models[1], kwargs) if (models[1] is not None and sigma0_cr is not None) else None
    if sigma0_cr is not None and lut_cr is None:
        raise ValueError("a cross-pol sigma0 was given but `model` names no cross-pol model")

    on_device = _device.any_device_array(inc, sigma0_co, sigma0_cr, None if np.isscalar(dsig_cr) else dsig_cr, ancillary_wind)
    if hasattr(want_codes, "begin") and not any(_is_xr(v) or _is_dask(v) for v in (inc, sigma0, sigma0_dual, ancillary_wind) if v is not None):
        # the gathered multi-GPU call: this rank's tile -> grid codes in device memory, chunk by chunk behind `want_codes` (the
        # pipeline's sink); numpy or device rasters.  Everything that can raise has happened when this returns.
        mode = "dual" if sigma0_dual is not None else ("mono_co" if models[0] is not None else "mono_cr")
        launch = _engine.invert_coded(lut_co, lut_cr, inc, sigma0_co, sigma0_cr, dsig_cr, None if no_ancillary else ancillary_wind,
                                      dsig_co, want_codes, dual_select=sigma0_dual is not None)
        return CodedWinds(mode, lut_co, lut_cr, None, None, launch=launch, on_device=on_device)
    if on_device:
        # rasters resident in HBM (torch CUDA tensors / __cuda_array_interface__): torch tensors on the same device come back,
        # nothing crosses PCIe; same routing and return conventions as below (:415-439), the dual-pol select fused in the kernel
        ws_co, ws_cr = _engine.invert_device(lut_co, lut_cr, inc, sigma0_co, sigma0_cr, dsig_cr,
                                             None if no_ancillary else ancillary_wind, dsig_co=dsig_co,
                                             dual_select=sigma0_dual is not None)
        logger.debug("timing invert_from_model (device rasters, asynchronous) : %.2fs.", time.time() - t0)
        if sigma0_dual is None:
            return ws_co if models[0] is not None else ws_cr.abs()
        return ws_co, ws_cr

    def _numpy(np_inc, np_co, np_cr, np_dsig, np_anc, codes=False):
        return _engine.invert_numpy(lut_co, lut_cr, np_inc, np_co, np_cr, np_dsig, np_anc, dsig_co=dsig_co, codes=codes)

    # cross-pol search disabled for every pixel when all cross sigma0 are NaN (:170) is implicit: NaN pixels skip it
    template = next((v for v in (sigma0, inc, sigma0_dual, ancillary_wind) if _is_xr(v)), None)
    args = (inc, sigma0_co, sigma0_cr, dsig_cr, None if no_ancillary else ancillary_wind)

    # dask blocks only ever run behind an xarray container, as in the reference: there `xr.zeros_like` raises TypeError for
    # anything that is not a DataArray (raw dask arrays included) and the whole call falls through to the numpy path,
    # which materialises its inputs and returns numpy (windspeed.py:337-386)
    if template is not None and any(_is_dask(v) for v in args if v is not None and not np.isscalar(v)):
        # dask in -> lazy dask out, one device call per row block (core dimension = last axis, :356-364)
        present = [i for i, v in enumerate(args) if v is not None and not np.isscalar(v)]

        def _block(*blocks):
            full = list(args)
            for i, b in zip(present, blocks):
                full[i] = b
            co_b, cr_b = _numpy(*full)
            nan_c = lambda: np.full(np.shape(full[0]), np.nan * 1j, dtype=np.complex128)
            return (co_b if co_b is not None else nan_c()), (cr_b if cr_b is not None else nan_c())

        ws_co, ws_cr = da.apply_gufunc(_block, ",".join(["(n)"] * len(present)) + "->(n),(n)",
                                       *[args[i].data if _is_xr(args[i]) else args[i] for i in present],
                                       output_dtypes=(np.complex128, np.complex128))
        if models[0] is None:
            ws_co = None
        if sigma0_cr is None:
            ws_cr = None
    else:
        np_args = [None if v is None else (v if np.isscalar(v) else np.asarray(v)) for v in args]
        if want_codes is True and template is None:
            mode = "dual" if sigma0_dual is not None else ("mono_co" if models[0] is not None else "mono_cr")
            return CodedWinds(mode, lut_co, lut_cr, *_numpy(*np_args, codes=True))
        ws_co, ws_cr = _numpy(*np_args)

    if template is not None:
        def wrap(values):
            if values is None:
                return None
            out = xr.zeros_like(template, dtype=np.complex128)
            out.data = values
            out.name = "windspeed_gmf"
            out.attrs.clear()
            return out
        ws_co, ws_cr = wrap(ws_co), wrap(ws_cr)

    logger.debug("timing invert_from_model : %.2fs.", time.time() - t0)

    if models[0] and models[0].iscopol and template is not None:
        ws_co.attrs["comment"] = f"wind speed and direction inverted from model {models[0].name} ({models[0].pol})"
        ws_co.attrs["model"] = models[0].name

    if sigma0_dual is None:
        if models[0] is not None:
            return ws_co  # mono co-pol
        ws = np.abs(ws_cr) if template is not None else _engine.abs_blocks(ws_cr)  # mono cross-pol: speed only
        if template is not None:
            ws.attrs["comment"] = f"wind speed inverted from model {models[1].name} ({models[1].pol})"
            ws.attrs["model"] = models[1].name
            ws.attrs["units"] = "m/s"
        return ws

    # dual-pol: keep the co-pol wind where either solution is below 5 m/s (:426-428)
    if template is not None:
        wspd_dual = xr.where((np.abs(ws_co) < 5) | (np.abs(ws_cr) < 5), ws_co, ws_cr)
        wspd_dual.attrs["comment"] = (f"wind speed and direction inverted from model {models[0].name} ({models[0].pol})"
                                      f" and {models[1].name} ({models[1].pol})")
        wspd_dual.attrs["model"] = f"{models[0].name} {models[1].name}"
    else:
        wspd_dual = _engine.dual_select(ws_co, ws_cr)
    return ws_co, wspd_dual
