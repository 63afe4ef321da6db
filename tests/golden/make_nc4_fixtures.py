"""Writes the netCDF-4 (HDF5) LUT fixtures of tests/golden/nc4/ -- run with an interpreter that has h5py (the build container's
/opt/conda/bin/python3.9: h5py 3.3.0 / HDF5 1.10.6; neither the product's interpreter nor the GPU image has any HDF5 package):

    /opt/conda/bin/python3.9 tests/golden/make_nc4_fixtures.py

The files follow the xsarsea LUT schema (reference: windspeed/models.py:232-262 `Model.to_netcdf`; read back by `NcLutModel`,
:350-410) in the container layouts the two netCDF-4 backends of xarray produce:
  * netCDF4-python / netCDF-C: creation order tracked and indexed on groups and attributes (hence DENSE attribute storage once
    the root group holds more than 8 attributes: it holds 11), fixed-length NC_CHAR text attributes, numeric attributes as
    1-element arrays, dimension scales (CLASS / NAME / _Netcdf4Dimid / REFERENCE_LIST, DIMENSION_LIST on the data variable),
    `_NCProperties`, `_FillValue` on the data variable; contiguous, or chunked + shuffle + deflate (+ fletcher32);
  * h5netcdf / h5py: variable-length UTF-8 string attributes, scalar numeric attributes, old-style groups and compact
    attributes (libver earliest, no creation order), or the latest file format (superblock 3, version-2 object headers).
Expected contents travel beside each file as <name>.expected.npz.  No file written by netCDF-C itself was available: the
structures are those HDF5 1.10.6 produces for the same property lists.
"""
import os

import h5py
import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nc4")


def table(n_inc, n_w, n_phi, seed):
    rng = np.random.default_rng(seed)
    inc = np.linspace(17.0, 50.0, n_inc)
    w = np.linspace(0.4, 0.4 * n_w, n_w)
    if n_phi:
        phi = np.linspace(0.0, 180.0, n_phi)
        v = -30 + 0.3 * inc[:, None, None] + 8 * np.log10(w)[None, :, None] + 1.5 * np.cos(np.radians(phi))[None, None, :]
        v = v + 0.01 * rng.standard_normal(v.shape)
        return inc, w, phi, v
    v = -40 + 0.1 * inc[:, None] + 10 * np.log10(w)[None, :] + 0.01 * rng.standard_normal((n_inc, n_w))
    return inc, w, None, v


def write(name, style, n_phi=5, chunked=False, fletcher=False, libver=None, dtype="<f8", seed=0, pol="VV", extra_attrs=0):
    inc, w, phi, v = table(6, 7, n_phi, seed)
    path = os.path.join(HERE, name)
    nc4 = style == "netcdf4"
    kw = {}
    if libver:
        kw["libver"] = libver
    with h5py.File(path, "w", track_order=nc4, **kw) as f:
        dims = [("incidence", inc), ("wspd", w)] + ([("phi", phi)] if phi is not None else [])
        for k, (dn, ax) in enumerate(dims):
            d = f.create_dataset(dn, data=ax.astype("<f8"), track_order=nc4)
            d.make_scale(dn)
            if nc4:
                d.attrs.create("_Netcdf4Dimid", np.int32(k))
        opts = dict(chunks=(3, 4, 2)[: v.ndim], shuffle=True, compression="gzip", compression_opts=4, fletcher32=fletcher) if chunked else {}
        s = f.create_dataset("sigma0_model", data=v.astype(dtype), track_order=nc4, **opts)
        for k, (dn, _) in enumerate(dims):
            s.dims[k].attach_scale(f[dn])
        s.attrs.create("_FillValue", np.array([np.nan], dtype=dtype) if nc4 else np.array(np.nan, dtype=dtype))
        steps = dict(inc_step=np.round(np.diff(inc)[0], 2), wspd_step=np.round(np.diff(w)[0], 2))
        ranges = dict(inc_range=[17.0, 50.0], wspd_range=[0.4, 0.4 * 7])
        if phi is not None:
            steps["phi_step"] = np.round(np.diff(phi)[0], 2)
            ranges["phi_range"] = [0.0, 180.0]
        text = dict(units="dB", resolution="low", model="cmod_fixture", pol=pol)
        if nc4:
            f.attrs.create("_NCProperties", np.bytes_("version=2,netcdf=4.7.4,hdf5=1.10.6"))
            for k in range(extra_attrs):  # a file with a long history: the attribute heap outgrows its first direct block
                f.attrs.create(f"history_{k:02d}", np.bytes_(f"step {k}: " + "processing note " * (3 + k % 5)))
            for k, val in text.items():
                f.attrs.create(k, np.bytes_(val))                       # NC_CHAR: fixed-length string, scalar dataspace
            for k, val in {**ranges, **steps}.items():
                f.attrs.create(k, np.atleast_1d(np.asarray(val, dtype="<f8")))   # NC_DOUBLE[n]
        else:
            for k, val in text.items():
                f.attrs[k] = val                                         # variable-length UTF-8 string
            for k, val in ranges.items():
                f.attrs[k] = np.asarray(val, dtype="<f8")
            for k, val in steps.items():
                f.attrs[k] = np.float64(val)                             # scalar dataspace
    exp = dict(values=v.astype(dtype).astype(np.float64), incidence=inc, wspd=w, **{k: np.asarray(x) for k, x in {**ranges, **steps}.items()},
               **{k: np.array(x) for k, x in text.items()})
    if phi is not None:
        exp["phi"] = phi
    np.savez(path + ".expected.npz", **exp)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    os.makedirs(HERE, exist_ok=True)
    write("nc_lut_netcdf4_contiguous.nc", "netcdf4", seed=1)
    write("nc_lut_netcdf4_deflate.nc", "netcdf4", chunked=True, seed=2)
    write("nc_lut_netcdf4_deflate_fletcher_f32.nc", "netcdf4", chunked=True, fletcher=True, dtype="<f4", seed=3)
    write("nc_lut_netcdf4_crosspol.nc", "netcdf4", n_phi=0, seed=4, pol="VH")
    write("nc_lut_h5netcdf_earliest.nc", "h5netcdf", seed=5)
    write("nc_lut_h5netcdf_deflate.nc", "h5netcdf", chunked=True, seed=6)
    write("nc_lut_h5netcdf_latest.nc", "h5netcdf", libver="latest", seed=7)
    write("nc_lut_netcdf4_v18.nc", "netcdf4", libver=("v108", "latest"), seed=8)
    write("nc_lut_netcdf4_many_attrs.nc", "netcdf4", seed=9, extra_attrs=18)
    write("nc_lut_h5py_latest_chunked.nc", "h5netcdf", chunked=True, libver="latest", seed=10)  # layout v4 chunk index: refused
