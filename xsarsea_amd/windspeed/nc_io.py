"""netCDF I/O of xsarsea-format LUT files (reference: windspeed/models.py:232-262 `Model.to_netcdf`, :350-410 `NcLutModel`)
without xarray: the classic netCDF-3 container through `scipy.io.netcdf_file`.

Schema (what the reference writes with `lut.to_dataset(promote_attrs=True).to_netcdf(file)`): dimensions / coordinate
variables `incidence`, `wspd` [, `phi`] (float64), data variable `sigma0_model` (float64, dB) over them, global attributes
`units`, `resolution`, `model` (short name), `pol`, `inc_range`, `wspd_range` [, `phi_range`], `inc_step`, `wspd_step`
[, `phi_step`].  xarray writes netCDF-4/HDF5 when the netCDF4 library is installed and classic netCDF-3 otherwise; this
module reads and writes the classic form (magic `CDF\\x01` / `CDF\\x02`).  An HDF5-based file (magic `\\x89HDF`) needs xarray
with a netCDF-4 backend, which neither the build nor the GPU image has.
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


def read_attrs(path):
    """Global attributes of a classic-netCDF LUT file as plain Python / numpy values."""
    from scipy.io import netcdf_file
    if not is_classic_netcdf(path):
        raise ImportError(f"{path} is not a classic netCDF-3 file (HDF5-based netCDF-4 needs xarray + netCDF4/h5netcdf)")
    with netcdf_file(path, "r", mmap=False) as f:
        return {k: _decode(v) for k, v in f._attributes.items()}


def read_lut(path):
    """-> Lut (values[incidence, wspd(, phi)] float64 + axes, attrs units / resolution / model from the global attributes)."""
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
