"""windspeed: wind retrieval from sigma0 and models (public names of `xsarsea.windspeed`)."""
__all__ = ["invert_from_model", "available_models", "get_model", "register_cmod7", "register_pickle_luts", "register_nc_luts",
           "register_luts", "nesz_flattening", "GmfModel", "Model", "gmfs", "gmfs_impl", "get_dsig", "get_dsig_wspd"]

from . import gmfs, gmfs_impl
from .cmod7 import register_cmod7
from .gmfs import GmfModel
from .models import Model, available_models, get_model, register_luts, register_nc_luts
from .pickle_luts import register_pickle_luts
from .utils import get_dsig, get_dsig_wspd, nesz_flattening
from .windspeed import invert_from_model
