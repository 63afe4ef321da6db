"""netCDF I/O of xsarsea-format LUT files (reference: windspeed/models.py:232-262 `Model.to_netcdf`, :350-410 `NcLutModel`)
without xarray: the classic netCDF-3 container through `scipy.io.netcdf_file`, and (reading) the HDF5 container of netCDF-4
files through the package's own minimal reader (`hdf5_min`: no HDF5 library exists in either image).

Schema (what the reference writes with `lut.to_dataset(promote_attrs=True).to_netcdf(file)`): dimensions / coordinate
variables `incidence`, `wspd` [, `phi`] (float64), data variable `sigma0_model` (float64, dB) over them, global attributes
`units`, `resolution`, `model` (short name), `pol`, `inc_range`, `wspd_range` [, `phi_range`], `inc_step`, `wspd_step`
[, `phi_step`].  xarray writes netCDF-4/HDF5 when the netCDF4 library (or h5netcdf) is installed and classic netCDF-3
otherwise; this module WRITES the classic form (magic `CDF\\x01` / `CDF\\x02`) and READS both: classic through scipy, HDF5-based
(magic `\\x89HDF`) through `hdf5_min` -- the layouts of both netCDF-4 backends (dense / compact attribute storage, fixed- and
variable-length text attributes, contiguous or chunked + shuffle + deflate + fletcher32 variables, superblocks 0-3).
"""
import numpy as np

from .lut import DIMS2, DIMS3, Lut

_TEXT_ATTRS = ("units", "resolution", "model", "pol")


def is_classic_netcdf(path):
    with open(path, "rb") as f:
        magic = f.read(4)
    return magic[:3] == b"CDF" and magic[3:4] in (b"\x01", b"\x02")


def _decode(v):
    if isinstance(v, bytes):
        return v.decode("utf-8")
    if isinstance(v, np.ndarray):
        if v.dtype.kind == "S":
            return v.tobytes().decode("utf-8")
        return v.item() if v.size == 1 else v.copy()
    return v


def _hdf5_attr(v):
    """hdf5_min attribute value -> what the classic route yields (str, float, or float array)."""
    if isinstance(v, str) or v is None:
        return v
    if isinstance(v, list):
        return v[0] if len(v) == 1 and isinstance(v[0], str) else v
    a = np.asarray(v)
    if a.dtype.kind in "fiu":
        return a.item() if a.size == 1 else a.astype(np.float64)
    return v


def read_attrs(path):
    """Global attributes of a LUT file (classic netCDF-3 or netCDF-4 / HDF5) as plain Python / numpy values."""
    from . import hdf5_min
    if hdf5_min.is_hdf5(path):
        return {k: _hdf5_attr(v) for k, v in hdf5_min.File(path).attrs.items() if not k.startswith("_NC")}
    from scipy.io import netcdf_file
    if not is_classic_netcdf(path):
        raise ImportError(f"{path} is neither a classic netCDF-3 file nor an HDF5-based netCDF-4 file")
    with netcdf_file(path, "r", mmap=False) as f:
        return {k: _decode(v) for k, v in f._attributes.items()}


def _read_lut_hdf5(path):
    from . import hdf5_min
    f = hdf5_min.File(path)
    attrs = {k: _hdf5_attr(v) for k, v in f.attrs.items()}
    if "sigma0_model" not in f.names():
        raise KeyError(f"no variable 'sigma0_model' in {path} (found {f.names()})")
    values = np.asarray(f.read("sigma0_model"), dtype=np.float64)
    dims = f.dims("sigma0_model")
    if dims is None or any(d is None for d in dims):  # no dimension scales attached: the schema's own order
        dims = DIMS3 if values.ndim == 3 else DIMS2
    dims = tuple(dims)
    if dims not in (DIMS2, DIMS3):
        raise IndexError(f"Bad dims '{dims}'. Should be '{DIMS2}' or '{DIMS3}'")
    axes = {d: np.asarray(f.read(d), dtype=np.float64) for d in dims}
    if f.undefined_fill_used:
        raise ValueError(f"{path}: part of the table (or of an axis) was never written and the file defines no fill value for it: "
                         "HDF5 would read zeros there, a plausible dB value -- refusing to search a table with holes")
    fill = f.dataset_attrs("sigma0_model").get("_FillValue")
    if fill is not None:  # xarray's mask_and_scale: _FillValue -> NaN (a no-op for the NaN fill xarray itself writes)
        fv = np.asarray(fill, dtype=np.float64).reshape(-1)[0]
        if fv == fv:
            values = np.where(values == fv, np.nan, values)
    if values.shape != tuple(len(axes[d]) for d in dims):
        raise ValueError(f"{path}: sigma0_model has shape {values.shape}, its coordinates {[len(axes[d]) for d in dims]}")
    return Lut(values, axes["incidence"], axes["wspd"], axes.get("phi"), units=attrs["units"], resolution=attrs["resolution"],
               model=attrs.get("model"))


def read_lut(path):
    """-> Lut (values[incidence, wspd(, phi)] float64 + axes, attrs units / resolution / model from the global attributes)."""
    from . import hdf5_min
    if hdf5_min.is_hdf5(path):
        return _read_lut_hdf5(path)
    from scipy.io import netcdf_file
    attrs = read_attrs(path)
    with netcdf_file(path, "r", mmap=False) as f:
        var = f.variables["sigma0_model"]
        dims = tuple(var.dimensions)
        if dims not in (DIMS2, DIMS3):
            raise IndexError(f"Bad dims '{dims}'. Should be '{DIMS2}' or '{DIMS3}'")
        values = np.array(var[:], dtype=np.float64)
        axes = {d: np.array(f.variables[d][:], dtype=np.float64) for d in dims}
    return Lut(values, axes["incidence"], axes["wspd"], axes.get("phi"), units=attrs["units"], resolution=attrs["resolution"],
               model=attrs.get("model"))


def write_lut(path, lut, global_attrs):
    """Write `lut` (a dB Lut) + global attributes in the reference's schema as netCDF-3 (64-bit offsets)."""
    from scipy.io import netcdf_file
    with netcdf_file(path, "w", version=2) as f:
        names = lut.dims
        for name, axis in zip(names, (lut.incidence, lut.wspd, lut.phi)):
            f.createDimension(name, len(axis))
            v = f.createVariable(name, "f8", (name,))
            v[:] = np.asarray(axis, dtype=np.float64)
        v = f.createVariable("sigma0_model", "f8", names)
        v[:] = lut.values
        for k, val in global_attrs.items():
            if val is None:
                continue
            if isinstance(val, str):
                setattr(f, k, val)
            else:
                setattr(f, k, np.atleast_1d(np.asarray(val, dtype=np.float64)) if np.ndim(val) else np.float64(val))
