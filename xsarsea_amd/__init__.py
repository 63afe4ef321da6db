"""xsarsea_amd: MI355X-native (gfx950) wind-inversion hot path of xsarsea.

Drop-in for `xsarsea.windspeed.invert_from_model` and `xsarsea.sigma0_detrend`: same signatures,
same container conventions, results identical to the reference's CPU path; the per-pixel work runs
in hand-written HIP kernels behind the C ABI of include/xsw.h (no CPU fallback).  See DESIGN.md.
"""
__version__ = "0.1.0"
__all__ = ["sigma0_detrend", "windspeed", "options", "dir_meteo_to_sample", "dir_sample_to_meteo", "dir_meteo_to_oceano",
           "dir_oceano_to_meteo", "dir_to_180", "dir_to_360", "read_sarwing_owi"]

from . import options, windspeed
from .detrend import (dir_meteo_to_oceano, dir_meteo_to_sample, dir_oceano_to_meteo, dir_sample_to_meteo, dir_to_180,
                      dir_to_360, read_sarwing_owi, sigma0_detrend)
