"""`sigma0_detrend`: drop-in for `xsarsea.sigma0_detrend` (reference: src/xsarsea/detrend.py:8-68).

out[l, s] = sigma0[l, s] / (g[s] / nanmean(g)),   g[s] = GMF(inc[line 0, s], wind_speed_gmf, wind_dir_gmf)

The GMF row (one value per sample) is evaluated on the host by the model; the per-pixel divide --
the only per-pixel work, purely HBM-bound -- runs on the device (`xsw_detrend`, include/xsw.h).
"""
import logging
import time

import numpy as np

from . import _device, _lib, options
from .windspeed.lut import xr
from .windspeed.models import get_model

logger = logging.getLogger("xsarsea")


def sigma0_detrend(sigma0, inc_angle, wind_speed_gmf=np.array([10.0]), wind_dir_gmf=np.array([45.0]),
                   model="gmf_cmod5n"):
    """Remove the incidence-angle trend of `sigma0` with a GMF evaluated at a fixed wind.

    Parameters
    ----------
    sigma0 : array (line, sample)
        linear sigma0 (xarray.DataArray or numpy)
    inc_angle : array (line, sample)
        incidence angle in degrees, same shape
    wind_speed_gmf, wind_dir_gmf : 0-D or size-1 arrays
        wind speed (m/s) and direction (deg, relative to antenna) fed to the GMF
    model : str | Model

    Returns
    -------
    detrended sigma0 (float64), same container type as `sigma0`
    """
    t0 = time.time()
    model = get_model(model)
    wind_speed_gmf, wind_dir_gmf = np.asarray(wind_speed_gmf), np.asarray(wind_dir_gmf)
    if wind_speed_gmf.ndim > 1 or wind_dir_gmf.ndim > 1:
        raise ValueError("wind_speed_gmf and wind_dir_gmf must be 0D or 1D")
    for var in (wind_speed_gmf, wind_dir_gmf):
        if var.ndim == 1 and var.size > 1:
            raise ValueError("wind_speed_gmf and wind_dir_gmf size must be 1 or 0")

    is_xr = xr is not None and isinstance(inc_angle, xr.DataArray)
    on_device = _device.is_device_array(sigma0)
    if _device.is_device_array(inc_angle):  # only the first line's incidence row is needed on the host (detrend.py:55)
        inc_row = _device.as_tensor(inc_angle, _device.device_of(inc_angle))[0].double().cpu().numpy()
    else:
        inc_row = np.asarray(inc_angle.isel(line=0) if is_xr else np.asarray(inc_angle)[0], dtype=np.float64)
    if hasattr(model, "_gmf_pyfunc_scalar"):
        g = np.asarray(model(inc_row, np.broadcast_to(wind_speed_gmf.reshape(-1)[:1], inc_row.shape),
                             np.broadcast_to(wind_dir_gmf.reshape(-1)[:1], inc_row.shape), broadcast=True),
                       dtype=np.float64)
    else:  # table model: (sample, 1, 1) grid, squeezed (detrend.py:57-61)
        g = np.asarray(model(inc_row, wind_speed_gmf.reshape(-1)[:1].astype(np.float64),
                             wind_dir_gmf.reshape(-1)[:1].astype(np.float64)), dtype=np.float64).reshape(inc_row.shape)
    ratio = g / np.nanmean(g)

    if on_device:  # sigma0 resident in HBM: a float64 torch tensor on the same device comes back, asynchronously
        import torch
        dev = _device.device_of(sigma0)
        t = _device.as_tensor(sigma0, dev)
        if t.dtype not in (torch.float32, torch.float64):
            t = t.double()
        t = t.contiguous()
        res = torch.empty(t.shape, dtype=torch.float64, device=dev)
        if t.numel():
            ctx = _lib.default_context(dev.index if dev.index is not None else torch.cuda.current_device())
            with _device.on_current_stream(ctx, dev):
                ctx.detrend_raw(t.numel() // t.shape[-1], t.shape[-1], _device.xsw_dtype(t), _lib.XSW_F64, _lib.MEM_DEVICE,
                                t.data_ptr(), ratio, res.data_ptr())
                t.record_stream(torch.cuda.current_stream(dev))
        return res
    values = np.asarray(sigma0)
    ctx = _lib.default_context(options.device)
    out = ctx.detrend_host(values.reshape(-1, values.shape[-1]), ratio).reshape(values.shape)
    logger.info("timing sigma0_detrend : %.2fs.", time.time() - t0)
    if xr is not None and isinstance(sigma0, xr.DataArray):
        res = sigma0.copy(data=out) if sigma0.dtype == out.dtype else sigma0.astype(np.float64).copy(data=out)
        res.attrs["comment"] = f"detrended with model {model.name}"
        return res
    return out


# ---- direction-convention helpers used by callers around the hot path (reference: detrend.py:96-201).
# One-line elementwise formulas; they turn the complex antenna-convention output into met/ocean directions.
def dir_meteo_to_sample(meteo_dir, ground_heading):
    """Meteorological direction (deg from north, clockwise) -> angle relative to the sample axis (rad, anticlockwise)."""
    return np.pi / 2 - np.deg2rad(meteo_dir - ground_heading)


def dir_sample_to_meteo(sample_dir, ground_heading):
    """Angle relative to the sample axis (deg, anticlockwise) -> meteorological direction (deg from north)."""
    return 90 - sample_dir + ground_heading


def dir_meteo_to_oceano(meteo_dir):
    """'from' convention -> 'to' convention (deg)."""
    return (meteo_dir + 180) % 360


def dir_oceano_to_meteo(oceano_dir):
    """'to' convention -> 'from' convention (deg)."""
    return (oceano_dir - 180) % 360


def dir_to_180(angle):
    """Wrap degrees into [-180, 180)."""
    return (angle + 180) % 360 - 180


def dir_to_360(angle):
    """Wrap degrees into [0, 360)."""
    return (angle + 360) % 360


def read_sarwing_owi(owi_file):
    """Open a sarwing OWI netCDF product in the xsar-like layout the notebooks use (needs xarray)."""
    from .windspeed.lut import xr as _xr
    if _xr is None:
        raise ImportError("read_sarwing_owi needs xarray")
    ds = _xr.merge([_xr.open_dataset(owi_file), _xr.open_dataset(owi_file, group="owiInversionTables_UV")])
    ds = ds.rename_dims({"owiAzSize": "line", "owiRaSize": "sample"}).drop_vars(["owiCalConstObsi", "owiCalConstInci"])
    return ds.assign_coords({"line": np.arange(len(ds.line)), "sample": np.arange(len(ds.sample))})
