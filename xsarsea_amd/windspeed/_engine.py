"""Glue between the model layer and libxsw: LUT upload cache and the numpy-in/numpy-out call that
stands where the reference's `_invert_from_model_numpy` stands (windspeed/windspeed.py:132-331)."""
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

    def build(self, ctx):
        target = dict(inc=self.incidence, wspd=self.wspd)
        if self.phi is not None:
            target.update(phi=self.phi, **host_tables(self.wspd, self.phi))
        ctx.build_lut(self.gmf_id, self.raw_axes, target)


_device_luts = {}


def lut_source(model, kwargs):
    """The dB LUT `invert_from_model` searches for `model`: the host-prepared `Lut` (`Model._lut`, memoised; the default:
    bit parity with a CPU run of the reference on this host), or -- `options.lut_build = "device"`, built-in GMFs only -- a
    `DeviceLut` whose table never exists on the host."""
    if options.lut_build == "device" and hasattr(model, "device_lut_plan"):
        plan = model.device_lut_plan(**kwargs)
        if plan is not None:
            key = (model.name,) + tuple(sorted(kwargs.items()))
            hit = _device_luts.get(key)
            if hit is None:
                hit = _device_luts[key] = DeviceLut(model.name, plan[0], plan[1], plan[2], key)
            return hit
    return model._lut(units="dB", **kwargs)


def ensure_luts(ctx, lut_co, lut_cr):
    """Upload (or build in place) the dB LUT objects unless this context already holds exactly them."""
    key_co, key_cr = ctx.lut_key
    up_co = lut_co is not None and key_co is not lut_co
    up_cr = lut_cr is not None and key_cr is not lut_cr
    for lut, up in ((lut_co, up_co), (lut_cr, up_cr)):
        if up and isinstance(lut, DeviceLut):
            lut.build(ctx)
    host_co = up_co and not isinstance(lut_co, DeviceLut)
    host_cr = up_cr and not isinstance(lut_cr, DeviceLut)
    if host_co or host_cr:
        ctx.upload_luts(co=_co_dict(lut_co) if host_co else None, cr=_cr_dict(lut_cr) if host_cr else None)
    if up_co or up_cr:
        ctx.lut_key = (lut_co if up_co else key_co, lut_cr if up_cr else key_cr)


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


def invert_numpy(lut_co, lut_cr, inc, sigma0_co, sigma0_cr, dsig_cr, anc, dsig_co=0.1):
    """(ws_co, ws_cr) complex128 for numpy rasters (None for a search that was not requested); any of
    sigma0_co / sigma0_cr / anc may be None.

    Raster dtypes follow the reference: the dB conversion runs in each sigma0's own dtype, then
    everything is handled as float64/complex128 (the gufunc signature, windspeed.py:308-318).  When
    every raster is float32/complex64 the device reads them as such (half the PCIe and HBM bytes)
    and widens in registers, which is the same arithmetic.
    """
    ctx = _lib.default_context(options.device)
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
    if is_db:
        sigma0_co = None if sigma0_co is None else _to_db(np.asarray(sigma0_co))
        sigma0_cr_lin = sigma0_cr
        sigma0_cr = None if sigma0_cr is None else _to_db(np.asarray(sigma0_cr))
    else:
        sigma0_cr_lin = sigma0_cr
    dt = np.float32 if all_f32 else np.float64
    if sigma0_cr is not None and np.isscalar(dsig_cr):
        if is_db:  # the kernel derives the broadcast from linear sigma0; do it here as the reference does
            with np.errstate(all="ignore"):
                dsig_cr = np.asarray(sigma0_cr_lin) * 0 + dsig_cr  # windspeed.py:122-123
        elif dt == np.float32:
            dsig_cr = float(np.float32(dsig_cr))
    cast = lambda a, t: None if a is None else np.ascontiguousarray(np.broadcast_to(np.asarray(a), shape), dtype=t)
    with ctx.lock:  # LUT upload + inversion as one step: another thread may want other LUTs on the same context
        ensure_luts(ctx, lut_co if sigma0_co is not None else None, lut_cr if sigma0_cr is not None else None)
        out_co, out_cr, _ = _invert(ctx, cast, dt, inc, sigma0_co, sigma0_cr, dsig_cr, anc, dsig_co, is_db)
    return out_co, out_cr  # None where that search did not run (the caller never reads it)


def _invert(ctx, cast, dt, inc, sigma0_co, sigma0_cr, dsig_cr, anc, dsig_co, is_db):
    return ctx.invert_host(
        cast(inc, dt), sigma0_co=cast(sigma0_co, dt), sigma0_cr=cast(sigma0_cr, dt),
        dsig_cr=dsig_cr if (dsig_cr is None or np.isscalar(dsig_cr)) else cast(dsig_cr, dt),
        anc=cast(anc, np.complex64 if dt == np.float32 else np.complex128), dsig_co=dsig_co, sigma0_is_db=is_db,
        algo=options.algo, out_dtype=np.complex128)
