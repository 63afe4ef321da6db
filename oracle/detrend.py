"""Oracle restatement of `sigma0_detrend` (reference: src/xsarsea/detrend.py:8-68), numpy inputs.
Test infrastructure only."""
import numpy as np

from . import gmf as _gmf


def sigma0_detrend(sigma0, inc_angle, wind_speed_gmf=np.array([10.0]), wind_dir_gmf=np.array([45.0]),
                   model="gmf_cmod5n"):
    if wind_speed_gmf.ndim > 1 or wind_dir_gmf.ndim > 1:  # detrend.py:36-40
        raise ValueError("wind_speed_gmf and wind_dir_gmf must be 0D or 1D")
    for var in (wind_speed_gmf, wind_dir_gmf):
        if var.ndim == 1 and var.size > 1:
            raise ValueError("wind_speed_gmf and wind_dir_gmf size must be 1 or 0")
    func = _gmf.GMFS[model][0]
    # model(inc[line 0], wspd, phi, broadcast=True) -> one value per sample (detrend.py:55-61)
    g = np.asarray(func(np.asarray(inc_angle)[0].astype(np.float64), np.asarray(wind_speed_gmf, dtype=np.float64),
                        np.asarray(wind_dir_gmf, dtype=np.float64)), dtype=np.float64)
    ratio = g / np.nanmean(g)  # :63
    return sigma0 / np.broadcast_to(ratio, np.shape(sigma0))  # :64
