"""Glue between the model layer and libxsw: LUT upload cache and the numpy-in/numpy-out call that
stands where the reference's `_invert_from_model_numpy` stands (windspeed/windspeed.py:132-331)."""
import ctypes

import numpy as np

from .. import _host, _lib, options


def host_tables(wspd, phi):
    """Tables of xsw_lut whose last bit depends on the math library, evaluated with numpy by the same
    expressions the reference uses (windspeed.py:167-168, :235-236, :257, :270-276), so that device
    results carry exactly the bits this host's CPU path would produce."""
    wspd = np.asarray(wspd, dtype=np.float64)
    phi = np.asarray(phi, dtype=np.float64)
    e = np.stack([np.exp(1j * np.deg2rad(phi)), np.exp(1j * np.deg2rad(-phi))])  # (2, n_phi)
    sol = wspd[None, :, None] * e[:, None, :]  # (2, n_wspd, n_phi)
    unit = np.exp(1j * np.angle(sol))
    return dict(cos_phi=np.cos(np.radians(phi)), sin_phi=np.sin(np.radians(phi)),
                out_dir=np.stack([e.real, e.imag], axis=-1), abs_co=np.abs(sol[0]),
                dual_dir=np.stack([unit.real, unit.imag], axis=-1))


def _ascending(values, axes):
    """The device wants strictly ascending axes (binary searches, uniform-grid windows); the reference takes any order
    (`np.argmin(abs(dim - x))` and a flat argmin over the table).  A LUT whose coordinate runs the other way (or is
    shuffled) is therefore re-ordered here, table permuted with it: the retrieved (wspd, phi) VALUES are those of the
    reference.  (Only the resolution of exact ties -- equal cost at two grid points -- follows the sorted order instead
    of the stored one.)  Repeated coordinates cannot be ordered and are refused."""
    values = np.asarray(values, dtype=np.float64)
    out_axes = []
    for k, ax in enumerate(axes):
        ax = np.asarray(ax, dtype=np.float64)
        if ax.size > 1 and not np.all(np.diff(ax) > 0):
            if np.isnan(ax).any() or np.unique(ax).size != ax.size:
                raise ValueError(f"LUT axis {k} holds NaN or repeated coordinates: cannot be inverted on the device")
            order = np.argsort(ax, kind="stable")
            ax = ax[order]
            values = np.take(values, order, axis=k)
        out_axes.append(ax)
    return np.ascontiguousarray(values), out_axes


def _co_dict(lut):
    db, (inc, wspd, phi) = _ascending(lut.values, (lut.incidence, lut.wspd, lut.phi))
    return dict(db=db, inc=inc, wspd=wspd, phi=phi, **host_tables(wspd, phi))


def _cr_dict(lut):
    db, (inc, wspd) = _ascending(lut.values, (lut.incidence, lut.wspd))
    return dict(db=db, inc=inc, wspd=wspd)


class DeviceLut:
    """A LUT that is BUILT on the device (`xsw_lut_build`: GMF grid fill -> interpolation -> dB -> search layout) instead of
    being prepared on the host and uploaded; stands where a host `Lut` stands in `invert_numpy`.  Carries the axes only."""

    def __init__(self, model_name, gmf_id, raw_axes, target_axes, key):
        self.model_name, self.gmf_id, self.raw_axes, self.key = model_name, gmf_id, raw_axes, key
        self.incidence, self.wspd, self.phi = target_axes
        self.shape = tuple(len(a) for a in target_axes if a is not None)

    def check_axes(self):
        for ax in (self.incidence, self.wspd, self.phi) + tuple(self.raw_axes):
            if ax is not None and np.size(ax) > 1 and not np.all(np.diff(np.asarray(ax, dtype=np.float64)) > 0):
                raise ValueError("device LUT build needs strictly ascending axes (the host route re-orders them: lut_build='host')")

    def build(self, ctx):
        self.check_axes()
        target = dict(inc=self.incidence, wspd=self.wspd)
        if self.phi is not None:
            target.update(phi=self.phi, **host_tables(self.wspd, self.phi))
        ctx.build_lut(self.gmf_id, self.raw_axes, target)


def lut_source(model, kwargs):
    """The dB LUT `invert_from_model` searches for `model`: the host-prepared `Lut` (`Model._lut`, memoised; the default:
    bit parity with a CPU run of the reference on this host), or -- `options.lut_build = "device"`, built-in GMFs only -- a
    `DeviceLut` whose table never exists on the host."""
    if options.lut_build == "device" and hasattr(model, "device_lut_plan"):
        plan = model.device_lut_plan(**kwargs)
        if plan is not None:
            # memoised on the model INSTANCE, like the host route's `Model._lut` (a model re-registered under the same name
            # with other ranges is another instance), and keyed by the plan's own axes, so a stale grid is never reused;
            # unhashable kwargs never reach a dict key
            key = (plan[0],) + tuple(None if a is None else np.asarray(a, dtype=np.float64).tobytes() for a in tuple(plan[1]) + tuple(plan[2]))
            cache = model.__dict__.setdefault("_device_luts", {})
            hit = cache.get(key)
            if hit is None:
                hit = cache[key] = DeviceLut(model.name, plan[0], plan[1], plan[2], key)
            return hit
    return model._lut(units="dB", **kwargs)


def ensure_luts(ctx, lut_co, lut_cr):
    """Upload (or build in place) the dB LUT objects unless this context already holds exactly them.  `ctx.lut_key` follows the
    context's tables step by step: a LUT is recorded the moment its install has succeeded, and a failing install leaves NO key
    for that slot (the context's table is then undefined: the next call installs again instead of searching a stale GMF)."""
    key_co, key_cr = ctx.lut_key
    up_co = lut_co is not None and key_co is not lut_co
    up_cr = lut_cr is not None and key_cr is not lut_cr
    for lut, up in ((lut_co, up_co), (lut_cr, up_cr)):  # validate both before the context is touched
        if up and isinstance(lut, DeviceLut):
            lut.check_axes()
    for slot, (lut, up) in enumerate(((lut_co, up_co), (lut_cr, up_cr))):
        if not up:
            continue
        key = list(ctx.lut_key)
        key[slot] = None
        ctx.lut_key = tuple(key)  # whatever happens below, the old table of this slot is gone
        if isinstance(lut, DeviceLut):
            lut.build(ctx)
        elif slot == 0:
            ctx.upload_luts(co=_co_dict(lut))
        else:
            ctx.upload_luts(cr=_cr_dict(lut))
        key[slot] = lut
        ctx.lut_key = tuple(key)


_BLOCK = _host.BLOCK
_pool = _host.pool


def _to_db(x):
    """10*log10(x + 1e-15) in x's dtype (windspeed.py:126-130).  Large rasters are cut into blocks handled by host
    threads: every element goes through the same numpy ufuncs, so the bits are those of the one-shot expression."""
    x = np.asarray(x)
    with np.errstate(all="ignore"):
        if x.size < 4 * _BLOCK or not x.flags.c_contiguous:
            return 10 * np.log10(x + 1e-15)
        out = np.empty(x.shape, dtype=(x[:1].ravel() + 1e-15).dtype)
        xf, of = x.reshape(-1), out.reshape(-1)

        def work(i):
            with np.errstate(all="ignore"):
                of[i:i + _BLOCK] = 10 * np.log10(xf[i:i + _BLOCK] + 1e-15)

        list(_pool().map(work, range(0, x.size, _BLOCK)))
        return out


def dual_select(ws_co, ws_cr):
    """np.where((np.abs(ws_co) < 5) | (np.abs(ws_cr) < 5), ws_co, ws_cr)  (windspeed.py:426-428), block-wise on the host
    pool for large rasters: the same numpy calls on every element, so the same bits (numpy's complex abs is not libm's
    hypot, which is why this select is not left to the device for numpy inputs)."""
    with np.errstate(all="ignore"):
        if ws_co.size < 4 * _BLOCK or not (ws_co.flags.c_contiguous and ws_cr.flags.c_contiguous):
            return np.where((np.abs(ws_co) < 5) | (np.abs(ws_cr) < 5), ws_co, ws_cr)
        out = _host.empty_touched(ws_co.shape, np.result_type(ws_co, ws_cr))
        a, b, o = ws_co.reshape(-1), ws_cr.reshape(-1), out.reshape(-1)

        def work(i):
            with np.errstate(all="ignore"):
                x, y = a[i:i + _BLOCK], b[i:i + _BLOCK]
                o[i:i + _BLOCK] = np.where((np.abs(x) < 5) | (np.abs(y) < 5), x, y)

        list(_pool().map(work, range(0, a.size, _BLOCK)))
        return out


def abs_blocks(z):
    """np.abs(z), block-wise on the host pool for large rasters (same bits)."""
    if z.size < 4 * _BLOCK or not z.flags.c_contiguous:
        return np.abs(z)
    out = _host.empty_touched(z.shape, np.abs(z.reshape(-1)[:1]).dtype)
    a, o = z.reshape(-1), out.reshape(-1)

    def work(i):
        with np.errstate(all="ignore"):
            o[i:i + _BLOCK] = np.abs(a[i:i + _BLOCK])

    list(_pool().map(work, range(0, a.size, _BLOCK)))
    return out


def any_valid(a):
    """`np.any(~np.isnan(a))` without materialising two rasters: block-wise with early exit."""
    a = np.asarray(a)
    if a.size <= _BLOCK or not a.flags.c_contiguous:
        return bool(np.any(~np.isnan(a)))
    flat = a.reshape(-1)
    return any(not np.isnan(flat[i:i + _BLOCK]).all() for i in range(0, a.size, _BLOCK))


def all_nan(a):
    """`np.all(np.isnan(a))`, block-wise with early exit."""
    return not any_valid(a)


def _device_list():
    """Devices of the single-process multi-GPU path (`options.devices`), or None for the one-device path."""
    devs = options.devices
    if devs is None:
        return None
    if isinstance(devs, str):
        if devs != "all":
            raise ValueError('options.devices must be None, "all" or a list of device indices')
        devs = list(range(_lib.device_count()))
    devs = [int(d) for d in devs]
    if not devs:
        raise ValueError("options.devices is an empty list")
    return devs  # (a one-entry list runs on THAT device: a single tile)


def tile_rows(lines, parts):
    """[(l0, l1)] contiguous row tiles: `lines // parts` lines each, the last one takes the remainder (multi_gpu.tile_bounds)."""
    base = lines // parts
    return [(k * base, lines if k == parts - 1 else (k + 1) * base) for k in range(parts)]


def invert_numpy(lut_co, lut_cr, inc, sigma0_co, sigma0_cr, dsig_cr, anc, dsig_co=0.1, codes=False):
    """(ws_co, ws_cr) complex128 for numpy rasters (None for a search that was not requested); any of
    sigma0_co / sigma0_cr / anc may be None.  codes=True: the uint32 grid codes instead (include/xsw.h: out_code_*; what
    `multi_gpu.invert_from_model_tiled` gathers -- 4 instead of 16 bytes per pixel -- and `expand_codes` turns into the winds).

    Raster dtypes follow the reference: the dB conversion runs in each sigma0's own dtype, then
    everything is handled as float64/complex128 (the gufunc signature, windspeed.py:308-318).  When
    every raster is float32/complex64 the device reads them as such (half the PCIe and HBM bytes)
    and widens in registers, which is the same arithmetic.

    `options.devices`: the raster's row tiles go to several GPUs from host threads, one libxsw context each, every tile
    written in place into the one output raster (pixels are independent: no exchange).
    """
    inc = np.asarray(inc)
    rasters = [a for a in (inc, sigma0_co, sigma0_cr, None if np.isscalar(dsig_cr) else dsig_cr) if a is not None]
    # the gufunc "(n),(n),(n),(n),(n)->(n),(n)" broadcasts its loop dimensions over ALL inputs (windspeed.py:307-322):
    # e.g. a 1-D incidence row with 2-D sigma0 gives (line, sample) outputs
    shape = np.broadcast_shapes(*(np.shape(a) for a in rasters + ([] if anc is None else [anc])))
    all_f32 = all(np.asarray(a).dtype == np.float32 for a in rasters) and (
        anc is None or np.asarray(anc).dtype == np.complex64)
    on_dev = options.db_on_device
    if on_dev == "auto":
        on_dev = not any(np.asarray(a).dtype == np.float32 for a in (sigma0_co, sigma0_cr) if a is not None)
    is_db = not on_dev
    dt = np.float32 if all_f32 else np.float64
    cast = lambda a, t: None if a is None else np.ascontiguousarray(np.broadcast_to(np.asarray(a), shape), dtype=t)
    cdt = np.complex64 if dt == np.float32 else np.complex128
    want_co, want_cr = sigma0_co is not None, sigma0_cr is not None
    lin = {}  # host dB: the linear sigma0 rasters in their OWN dtype; converted piece by piece inside the library's pipeline
    if is_db:
        # numpy's own log10 in the raster's dtype is the reference's arithmetic (windspeed.py:126-130) and, for float32, the
        # only way to its bits (a platform-specific few-ulp SIMD routine).  It runs as the STAGING step of xsw_invert's host
        # pipeline (xsw_invert_args.stage): the worker thread that is about to upload a piece calls back, numpy converts that
        # piece straight into the page-locked staging buffer -- no pass of its own, no dB raster in host memory.
        as_src = lambda a: np.ascontiguousarray(np.broadcast_to(np.asarray(a), shape))
        if want_co:
            lin[_lib.STAGE_SIGMA0_CO] = as_src(sigma0_co)
        if want_cr:
            lin[_lib.STAGE_SIGMA0_CR] = as_src(sigma0_cr)
    dsig_fill = None
    if want_cr and np.isscalar(dsig_cr):
        if is_db:  # the kernel derives the broadcast from linear sigma0; with dB rasters it is formed as the reference does
            dsig_fill = dsig_cr  # sigma0_cr * 0 + dsig_cr  (windspeed.py:122-123), piece by piece in the staging callback
        elif dt == np.float32:
            dsig_cr = float(np.float32(dsig_cr))
    full = dict(inc=cast(inc, dt), anc=cast(anc, cdt))
    # rasters the staging callback fills are never read through their pointer: the incidence raster stands in (right size)
    full["sigma0_co"] = None if not want_co else (full["inc"] if is_db else cast(sigma0_co, dt))
    full["sigma0_cr"] = None if not want_cr else (full["inc"] if is_db else cast(sigma0_cr, dt))
    if dsig_fill is not None:
        full["dsig_cr"] = full["inc"]
    else:
        full["dsig_cr"] = dsig_cr if (dsig_cr is None or np.isscalar(dsig_cr)) else cast(dsig_cr, dt)

    def stage_for(rows):
        """The staging callback of one row tile (pixel offsets are tile-local)."""
        if not lin:
            return None
        sl = (lambda a: a) if rows is None else (lambda a: a[rows[0]:rows[1]])
        src = {k: sl(v).reshape(-1) for k, v in lin.items()}
        item = np.dtype(dt).itemsize

        def stage(which, px0, npx, dst):
            if which == _lib.STAGE_DSIG_CR and dsig_fill is not None:
                x = src[_lib.STAGE_SIGMA0_CR][px0:px0 + npx]
            elif which in src:
                x = src[which][px0:px0 + npx]
            else:
                return 0
            out = np.frombuffer((ctypes.c_char * (npx * item)).from_address(dst), dtype=dt)
            with np.errstate(all="ignore"):
                if which == _lib.STAGE_DSIG_CR:
                    out[...] = x * 0 + dsig_fill
                elif x.dtype == dt:  # 10 * np.log10(x + 1e-15), the three ufunc loops writing in place
                    np.add(x, 1e-15, out=out)
                    np.log10(out, out=out)
                    np.multiply(out, 10, out=out)
                else:
                    out[...] = 10 * np.log10(x + 1e-15)
            return 1
        return stage

    def run(ctx, rows, out_co, out_cr):
        """One context inverts rows [l0, l1) of the (lines, samples) view of every raster, into the same rows of the outputs."""
        sl = (lambda a: a) if rows is None else (lambda a: a[rows[0]:rows[1]])
        with ctx.lock:  # LUT upload + inversion as one step: another thread may want other LUTs on the same context
            ensure_luts(ctx, lut_co if want_co else None, lut_cr if want_cr else None)
            if options.host_threads:
                ctx.set_host_threads(options.host_threads)
            res = ctx.invert_host(sl(full["inc"]), sigma0_co=None if not want_co else sl(full["sigma0_co"]),
                                   sigma0_cr=None if not want_cr else sl(full["sigma0_cr"]),
                                   dsig_cr=full["dsig_cr"] if (full["dsig_cr"] is None or np.isscalar(full["dsig_cr"])) else sl(full["dsig_cr"]),
                                   anc=None if full["anc"] is None else sl(full["anc"]), dsig_co=dsig_co, sigma0_is_db=is_db,
                                   algo=options.algo, out_dtype=np.complex128, out_co=None if out_co is None else sl(out_co),
                                   out_cr=None if out_cr is None else sl(out_cr), stage=stage_for(rows), want_codes=codes,
                                   want_complex=not codes)
            return (res[3][0], res[3][1], None) if codes else res

    devs = _device_list()
    n = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
    if codes or devs is None or len(devs) == 1 or len(shape) < 2 or n < options.devices_min_pixels or shape[0] < 4 * len(devs):
        # one device: `options.device`, or the one entry of `options.devices` (a one-entry list names THE device, whatever the
        # raster's size); several entries with a raster too small to tile: the first of them
        one = options.device if devs is None else devs[0]
        out_co, out_cr, _ = run(_lib.default_context(one), None, None, None)
        return out_co, out_cr  # None where that search did not run (the caller never reads it)
    # several GPUs: contiguous row tiles of the leading axis, one host thread and one context per GPU, results in place
    lines = shape[0]
    out_co = np.empty(shape, np.complex128) if want_co else None
    out_cr = np.empty(shape, np.complex128) if want_cr else None
    ctxs = _lib.contexts_for(devs)
    tiles = tile_rows(lines, len(ctxs))
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(ctxs)) as ex:
        futs = [ex.submit(run, c, t, out_co, out_cr) for c, t in zip(ctxs, tiles) if t[1] > t[0]]
        for f in futs:
            f.result()  # re-raises a tile's error here
    return out_co, out_cr


def expand_codes(lut_co, lut_cr, codes_co, codes_cr):
    """Grid codes -> (ws_co, ws_cr) complex128 on the host, from the tables of these LUTs (installed on the context if they are not)."""
    ctx = _lib.default_context(options.device)
    with ctx.lock:
        ensure_luts(ctx, lut_co if codes_co is not None else None, lut_cr if codes_cr is not None else None)
        return ctx.expand_codes_host(codes_co, codes_cr)


def invert_device(lut_co, lut_cr, inc, sigma0_co, sigma0_cr, dsig_cr, anc, dsig_co=0.1, dual_select=False):
    """(ws_co, ws_cr) torch complex tensors for rasters resident in HBM (torch CUDA tensors / `__cuda_array_interface__`
    objects; host arrays among them are uploaded): the drop-in call without PCIe.  sigma0 -> dB is fused into the kernel
    (float32 rasters: float32 arithmetic like the reference, the log10 correctly rounded -- numpy's float32 log10 is a few-ulp
    SIMD routine, so ~2e-5 of the pixels of a float32 raster land one grid step from a numpy run; float64 rasters agree bit for
    bit).  Asynchronous on torch's current stream.  dual_select: ws_cr receives the fused where(|co|<5 | |dual|<5, co, dual)."""
    import torch
    from .. import _device
    arrays = [a for a in (inc, sigma0_co, sigma0_cr, None if np.isscalar(dsig_cr) else dsig_cr, anc) if a is not None]
    dev = _device.device_of(*arrays)
    ctx = _lib.default_context(dev.index if dev.index is not None else torch.cuda.current_device())
    t_inc = _device.as_tensor(inc, dev)
    t_co = None if sigma0_co is None else _device.as_tensor(sigma0_co, dev)
    t_cr = None if sigma0_cr is None else _device.as_tensor(sigma0_cr, dev)
    t_dsig = None if (dsig_cr is None or np.isscalar(dsig_cr)) else _device.as_tensor(dsig_cr, dev)
    t_anc = None if anc is None else _device.as_tensor(anc, dev)
    rasters = [t for t in (t_inc, t_co, t_cr, t_dsig) if t is not None]
    shape = torch.broadcast_shapes(*(t.shape for t in rasters + ([] if t_anc is None else [t_anc])))
    all_f32 = all(t.dtype == torch.float32 for t in rasters) and (t_anc is None or t_anc.dtype == torch.complex64)
    rt, ct = (torch.float32, torch.complex64) if all_f32 else (torch.float64, torch.complex128)
    is_db = False
    if not all_f32 and any(t is not None and t.dtype == torch.float32 for t in (t_co, t_cr)):
        # mixed dtypes (float32 sigma0 next to a float64 incidence, say): the reference converts sigma0 to dB in sigma0's OWN dtype
        # (windspeed.py:126-130) before anything is widened -- do that here, then hand dB rasters to the kernel
        to_db = lambda t: None if t is None else (10 * torch.log10(t + 1e-15))
        # a scalar dsig_cr is broadcast from the LINEAR sigma0_cr, in its dtype (windspeed.py:122-123: sigma0_cr * 0 + dsig_cr --
        # finite where sigma0 is; formed from the dB value it would be NaN wherever sigma0_cr + 1e-15 == 0, i.e. -inf dB)
        if t_cr is not None and t_dsig is None and dsig_cr is not None:
            t_dsig = t_cr * 0 + dsig_cr
        t_co, t_cr, is_db = to_db(t_co), to_db(t_cr), True
    prep = lambda t, d: None if t is None else t.to(d).expand(shape).contiguous()
    t_inc, t_co, t_cr, t_dsig, t_anc = prep(t_inc, rt), prep(t_co, rt), prep(t_cr, rt), prep(t_dsig, rt), prep(t_anc, ct)
    dsig_scalar = 0.1
    if t_cr is not None and t_dsig is None and dsig_cr is not None:
        dsig_scalar = float(np.float32(dsig_cr)) if all_f32 else float(dsig_cr)
    odt = torch.complex64 if options.device_out_dtype == "complex64" else torch.complex128
    out_co = torch.empty(shape, dtype=odt, device=dev) if t_co is not None else None
    out_cr = torch.empty(shape, dtype=odt, device=dev) if t_cr is not None else None
    n = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
    lines, samples = (n // shape[-1], shape[-1]) if len(shape) and n else (1 if n else 0, 1 if n else 0)
    p = lambda t: None if t is None else t.data_ptr()
    if n:
        with _device.on_current_stream(ctx, dev):
            ensure_luts(ctx, lut_co if t_co is not None else None, lut_cr if t_cr is not None else None)
            ctx.invert_raw(lines, samples, _lib.XSW_F32 if all_f32 else _lib.XSW_F64,
                           _lib.XSW_F32 if odt == torch.complex64 else _lib.XSW_F64, _lib.MEM_DEVICE, p(t_inc), p(t_co), p(t_cr),
                           p(t_dsig), p(t_anc), p(out_co), p(out_cr), None, dsig_co, dsig_scalar, is_db,
                           _lib.ALGOS.get(options.algo, options.algo), dual_select and out_cr is not None and out_co is not None)
            # the inputs must outlive the asynchronous launch: tie them to the stream they are read on
            for t in (t_inc, t_co, t_cr, t_dsig, t_anc):
                if t is not None:
                    t.record_stream(torch.cuda.current_stream(dev))
    return out_co, out_cr


def invert_coded(lut_co, lut_cr, inc, sigma0_co, sigma0_cr, dsig_cr, anc, dsig_co, sink, dual_select=False):
    """A rank's row tile -> 4-byte grid codes in DEVICE memory, chunk by chunk, for the gathered multi-GPU call
    (`multi_gpu.invert_from_model_tiled`; `sink` is its `_CodeSink`: it owns the code buffers, starts the gather of every
    chunk and queues its expansion on the destination rank).  numpy or device rasters; nothing of the answer visits the host.

    Two phases, so that a rank that fails leaves no peer waiting in a transfer: everything that can raise for a reason of
    the inputs -- dtypes, shapes, LUT install, uploads -- happens HERE, before any exchange; the returned `launch()` only queues
    kernels and transfers (the caller agrees with the other ranks in between).

    numpy rasters follow `invert_numpy`'s arithmetic: float32 sigma0 is converted to dB on the host with numpy's own log10
    (`options.db_on_device = "auto"`; bit parity with the reference) -- through libxsw's staging ring while the other rasters
    are already resident (XSW_MEM_DEVICE_SIGMA0_HOST) -- and a scalar `dsig_cr` is broadcast from the LINEAR sigma0
    (windspeed.py:122-123).  Device rasters follow `invert_device` (sigma0 -> dB fused, the dual-pol select fused)."""
    import torch
    from .. import _device
    given = [a for a in (inc, sigma0_co, sigma0_cr, None if np.isscalar(dsig_cr) else dsig_cr, anc) if a is not None]
    on_device = _device.any_device_array(*given)
    dev = _device.device_of(*given) if on_device else torch.device("cuda", int(options.device))
    ctx = _lib.default_context(dev.index if dev.index is not None else torch.cuda.current_device())
    want_co, want_cr = sigma0_co is not None, sigma0_cr is not None
    shape = tuple(np.broadcast_shapes(*(tuple(np.shape(a)) for a in given)))
    n = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
    host_lin, is_db = {}, False
    if on_device:
        t = {k: (None if v is None or np.isscalar(v) else _device.as_tensor(v, dev)) for k, v in
             (("inc", inc), ("co", sigma0_co), ("cr", sigma0_cr), ("dsig", dsig_cr), ("anc", anc))}
        rasters = [t[k] for k in ("inc", "co", "cr", "dsig") if t[k] is not None]
        all_f32 = all(x.dtype == torch.float32 for x in rasters) and (t["anc"] is None or t["anc"].dtype == torch.complex64)
        if not all_f32 and any(t[k] is not None and t[k].dtype == torch.float32 for k in ("co", "cr")):
            if want_cr and t["dsig"] is None and dsig_cr is not None:  # scalar dsig_cr: broadcast from the LINEAR sigma0, in its dtype
                t["dsig"] = t["cr"] * 0 + dsig_cr
            to_db = lambda x: None if x is None else (10 * torch.log10(x + 1e-15))
            t["co"], t["cr"], is_db = to_db(t["co"]), to_db(t["cr"]), True
    else:
        arr = lambda a: None if a is None or np.isscalar(a) else np.asarray(a)
        h = dict(inc=arr(inc), co=arr(sigma0_co), cr=arr(sigma0_cr), dsig=arr(dsig_cr), anc=arr(anc))
        rasters = [h[k] for k in ("inc", "co", "cr", "dsig") if h[k] is not None]
        all_f32 = all(x.dtype == np.float32 for x in rasters) and (h["anc"] is None or h["anc"].dtype == np.complex64)
        on_dev = options.db_on_device
        if on_dev == "auto":
            on_dev = not any(h[k] is not None and h[k].dtype == np.float32 for k in ("co", "cr"))
        is_db = not on_dev
        if is_db and want_cr and h["dsig"] is None and dsig_cr is not None:
            with np.errstate(all="ignore"):
                h["dsig"] = h["cr"] * 0 + dsig_cr  # windspeed.py:122-123, from the linear sigma0 in its own dtype
        if is_db:  # sigma0 stays on the host: converted piece by piece on its way up
            host_lin = {k: np.ascontiguousarray(np.broadcast_to(h[k], shape)) for k in ("co", "cr") if h[k] is not None}
        t = {k: (None if v is None or (is_db and k in ("co", "cr")) else torch.from_numpy(np.ascontiguousarray(v)).to(dev)) for k, v in h.items()}
    rt, ct = (torch.float32, torch.complex64) if all_f32 else (torch.float64, torch.complex128)
    npdt = np.float32 if all_f32 else np.float64
    prep = lambda x, d: None if x is None else x.to(d).expand(shape).contiguous()
    t_inc, t_co, t_cr, t_dsig, t_anc = prep(t["inc"], rt), prep(t["co"], rt), prep(t["cr"], rt), prep(t["dsig"], rt), prep(t["anc"], ct)
    dsig_scalar = 0.1
    if want_cr and t_dsig is None and dsig_cr is not None:
        dsig_scalar = float(np.float32(dsig_cr)) if all_f32 else float(dsig_cr)
    numpy_out = not on_device
    odt = torch.complex128 if (numpy_out or options.device_out_dtype != "complex64") else torch.complex64
    sink.begin(shape, want_co, want_cr, dev, odt)
    pipe = sink.pipe
    S = pipe.samples
    item, oitem = (4 if all_f32 else 8), (8 if odt == torch.complex64 else 16)
    xdt, xodt = (_lib.XSW_F32 if all_f32 else _lib.XSW_F64), (_lib.XSW_F32 if odt == torch.complex64 else _lib.XSW_F64)
    algo = _lib.ALGOS.get(options.algo, options.algo)
    fused_select = bool(dual_select and want_co and want_cr and on_device)
    with ctx.lock:
        ensure_luts(ctx, lut_co if want_co else None, lut_cr if want_cr else None)  # (also on a rank whose tile is empty: it may expand)
    at = lambda x, off, size: None if x is None else x.data_ptr() + off * size
    src = {k: v.reshape(-1) for k, v in host_lin.items()}

    def stage_for(px_base):
        def stage(which, px0, npx, dst):
            key = "co" if which == _lib.STAGE_SIGMA0_CO else ("cr" if which == _lib.STAGE_SIGMA0_CR else None)
            if key not in src:
                return 0
            x = src[key][px_base + px0:px_base + px0 + npx]
            out = np.frombuffer((ctypes.c_char * (npx * item)).from_address(dst), dtype=npdt)
            with np.errstate(all="ignore"):
                if x.dtype == npdt:
                    np.add(x, 1e-15, out=out)
                    np.log10(out, out=out)
                    np.multiply(out, 10, out=out)
                else:
                    out[...] = 10 * np.log10(x + 1e-15)
            return 1
        return stage

    def invert_chunk(k, r0, r1):
        off, npx = r0 * S, (r1 - r0) * S
        lines, samples = ((r1 - r0), S) if len(shape) >= 2 else (1, npx)
        p_co = at(t_co, off, item) if t_co is not None else (at(t_inc, off, item) if "co" in src else None)
        p_cr = at(t_cr, off, item) if t_cr is not None else (at(t_inc, off, item) if "cr" in src else None)
        host_route = bool(src) and lines >= 4
        if src and not host_route:  # a chunk too thin for the staging ring: numpy's dB of these rows, uploaded
            with np.errstate(all="ignore"):
                up = {key: torch.from_numpy(np.ascontiguousarray(10 * np.log10(v[off:off + npx] + 1e-15)).astype(npdt, copy=False)).to(dev) for key, v in src.items()}
            p_co = up["co"].data_ptr() if "co" in up else p_co
            p_cr = up["cr"].data_ptr() if "cr" in up else p_cr
            for x in up.values():
                x.record_stream(torch.cuda.current_stream(dev))
        if host_route:
            # (pointers of the host rasters are never read: the staging callback fills every piece)
            ctx.invert_raw(lines, samples, xdt, xodt, _lib.MEM_DEVICE_SIGMA0_HOST, at(t_inc, off, item),
                           src["co"].ctypes.data + off * src["co"].itemsize if "co" in src else None,
                           src["cr"].ctypes.data + off * src["cr"].itemsize if "cr" in src else None,
                           at(t_dsig, off, item), at(t_anc, off, 2 * item), None, None, None, dsig_co, dsig_scalar, True, algo, False,
                           out_code_co=at(pipe.codes, off, 4), out_code_cr=at(pipe.codes_dual, off, 4), stage=stage_for(off))
        else:
            ctx.invert_raw(lines, samples, xdt, xodt, _lib.MEM_DEVICE, at(t_inc, off, item), p_co, p_cr, at(t_dsig, off, item),
                           at(t_anc, off, 2 * item), None, None, None, dsig_co, dsig_scalar, is_db, algo, fused_select,
                           out_code_co=at(pipe.codes, off, 4), out_code_cr=at(pipe.codes_dual, off, 4))

    def expand_rows(g0, g1, stream):
        off = g0 * S
        ctx.expand_codes_on_stream(stream.cuda_stream, (g1 - g0) * S, xodt, at(pipe.full_codes, off, 4), at(pipe.full_codes_dual, off, 4),
                                   at(pipe.full, off, oitem), at(pipe.full_dual, off, oitem))

    def launch():
        with _device.on_current_stream(ctx, dev):
            pipe.run(invert_chunk, expand_rows)
            for x in (t_inc, t_co, t_cr, t_dsig, t_anc):
                if x is not None:
                    x.record_stream(torch.cuda.current_stream(dev))

    return launch
