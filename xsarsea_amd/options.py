"""Run-time switches of the device path (module attributes; set them before calling the API)."""

#: HIP device index used by the drop-in API (one process per GPU: set it to LOCAL_RANK).
device = 0

#: Single-process multi-GPU for HOST rasters (numpy / xarray / dask blocks): None = `device` only; "all" = every visible GPU;
#: or a list of device indices.  The raster is cut into contiguous row tiles (the reference's dask strategy,
#: windspeed.py:356-364), one host thread and one libxsw context per GPU, every tile written in place into the one output
#: raster -- no collective.  This is what makes `invert_from_model` a drop-in for the reference's own in-call parallelism
#: (numba `target="parallel"` threads, windspeed.py:306-323).  Rasters below `devices_min_pixels` stay on `device`.
devices = None
devices_min_pixels = 1 << 22

#: complex dtype of the winds returned for DEVICE-resident inputs (torch CUDA tensors): "complex128" (the reference's
#: result dtype) or "complex64" (half the HBM bytes written).
device_out_dtype = "complex128"

#: worker threads of the host-memory paths of libxsw per context (0 = the library default: XSW_HOST_THREADS or 12).  Each worker
#: keeps a page-locked staging buffer (~40 MB for float32 mono rasters, ~110 MB for float64 dual-pol) and a device buffer (the
#: same + the chunk's work lists and records: ~90 MB) of one ~2 Mpx chunk between calls, per context (and per GPU with `devices`);
#: what exceeds XSW_STAGING_KEEP_MB (environment, default 1536, a worker counted with the larger of its two buffers) is released
#: after every call, lowering `host_threads` frees the surplus workers.  Device rasters: the context's work lists take 28.8 bytes
#: per pixel of the largest raster inverted so far (include/xsw.h: xsw_invert).
host_threads = 0

#: sigma0 -> dB conversion (windspeed.py:126-130).
#:   "auto"  : float64 rasters are converted on the device; float32 rasters on the host with numpy,
#:             because numpy's float32 log10 is a platform-specific few-ulp SIMD routine and only the
#:             host can reproduce its bits (DESIGN.md "float32 dB").
#:   True    : always on the device (fastest; float32 log10 correctly rounded instead of numpy's).
#:   False   : always on the host.
db_on_device = "auto"

#: search kernel: "auto" | "pruned" | "exhaustive" | "exact"  (include/xsw.h XSW_ALGO_*)
algo = "auto"

#: LUT resolution change (Model._normalize_lut): "auto" = on the device when one is present (bit-identical to
#: the host numpy path, ~100x faster at the default 501x499x181 size), "host" = numpy, "device" = always device.
lut_interp = "auto"

#: forward GMF of the built-in models on broadcast arrays (GmfModel.__call__(..., broadcast=True)):
#: "auto" = device for >= gmf_device_min_size elements when a device is present (values agree with the host
#: evaluation to ~1e-14 relative, not bit for bit; LUT preparation always evaluates on the host), "host", "device".
gmf_on_device = "auto"
gmf_device_min_size = 1 << 18

#: cross-pol noise flattening (`windspeed.nesz_flattening`): "auto" = device (`xsw_nesz_flatten`) for FLOAT64 rasters of
#: >= nesz_device_min_size pixels when a device is present -- float64 accumulation and closed-form least squares, within
#: 1e-10 relative of the host route, which reproduces numpy's polyfit bit for bit; float32 rasters stay on the host under
#: "auto" (the reference accumulates them in float32: the device's float64 sums differ from that by ~1e-6) -- "host",
#: "device" (every float raster; float32 within 1e-5).
nesz_on_device = "auto"
nesz_device_min_size = 1 << 20

#: where the dB LUT of a built-in GMF is prepared for `invert_from_model`:
#:   "host"   : `Model.to_lut` on the host (numpy GMF fill + interpolation (see lut_interp) + numpy log10), then uploaded --
#:              the table is bit-identical to what a CPU run of the reference on this host searches (the default);
#:   "device" : `xsw_lut_build` -- grid fill, interpolation, dB conversion and search layout all on the device, the
#:              362 MB table never exists on the host (values within ~1e-13 dB of the host-built table: device libm).
lut_build = "host"
