"""Cross-pol preprocessing helpers used right before `invert_from_model`
(reference: src/xsarsea/windspeed/utils.py): `get_dsig` (:47-91), `get_dsig_wspd` (:18-44),
`nesz_flattening` (:94-163).

`get_dsig*` are elementwise formulas (host numpy, bit-identical to the reference incl. its dtype promotion:
tests/golden/crosspol_prep.npz).  `nesz_flattening` is a full-raster pass -- column nan-mean, then one degree-1
least-squares fit per line in dB -- and runs on the device for large rasters (`xsw_nesz_flatten`, include/xsw.h;
`options.nesz_on_device`); the host route below reproduces the reference bit for bit.
"""
import warnings

import numpy as np

_DSIG_WSPD = {
    "dsig_wspd_rs2_v3": (-0.4908643753212401, 16.763199934792965, 1.3891445172991084, 20.616914824394343),
    "dsig_wspd_s1_ew_rec_v3": (-0.5858970325653666, 16.50039320910609, 1.1032031322520397, 7.434663633997121),
    "dsig_wspd_rcm_v3": (-0.7920301376936547, 15.8288289109038, 0.24040294696606557, 0.2538177092195224),
}
# logistic exponent c(inc) of the S1 v2 rule: rate, centre, floor, span.  numpy float64 scalars on purpose: like the
# reference's coefficient array they promote a float32 incidence raster to float64.
_S1_V2_EXPONENT = np.array([1.57952257, 25.61843791, 1.46852088, 1.4058646])


def get_dsig_wspd(name, U_crosspol, SNR_cr):
    """Weight alpha(U, SNR) in [0, 1]: logistic in (U - c0 + gamma*SNR) times a roll-off above Umax = 30."""
    b, c0_base, gamma, k = _DSIG_WSPD[name]
    centre = c0_base - gamma * SNR_cr
    core = 1 / (1 + np.exp(-b * (U_crosspol - centre)))
    drop = 1 / (1 + np.exp((U_crosspol - 30) * k))
    return np.clip(core * drop, 0, 1)


def get_dsig(name, inc, sigma0_cr, nesz_cr):
    """`dsig_cr` for `invert_from_model` from the cross-pol signal-to-noise ratio."""
    if name == "gmf_s1_v2":
        rate, centre, floor, span = _S1_V2_EXPONENT
        c = floor + span / (1 + np.exp(-rate * (inc - centre)))
        return 1 / np.sqrt(1 * (sigma0_cr / nesz_cr) ** c)
    if name == "gmf_rs2_v2":
        return 1 / np.sqrt(1 * (sigma0_cr / nesz_cr) ** 8)
    if name in ("sarwing_lut_cmodms1ahw", "nc_lut_cmodms1ahw"):
        return (1.25 / (sigma0_cr / nesz_cr)) ** 4.0
    raise ValueError("dsig names different than 'gmf_s1_v2' or 'gmf_rs2_v2' or 'sarwing_lut_cmodms1ahw' or "
                     "'nc_lut_cmodms1ahw' are not handled. You can compute your own dsig_cr.")


def _nesz_flattening_host(values, inc):
    """The reference's arithmetic, line by line (utils.py:119-163), dtypes left as they come."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        col_mean = np.nanmean(values, axis=0)
        inc_row = np.nanmean(inc, axis=0)  # "incidence is almost constant along line dim"
    out = np.empty(values.shape, dtype=np.float64)
    for i, row in enumerate(values):
        filled = row.copy()
        gap = np.isnan(filled)
        filled[gap] = col_mean[gap]
        with np.errstate(all="ignore"):
            db = 10.0 * np.log10(filled)
        ok = np.isfinite(db)
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                slope, icpt = np.polyfit(inc_row[ok], db[ok], 1)
        except TypeError:  # nothing to fit on this line
            out[i] = np.nan
            continue
        out[i] = 10.0 ** ((inc_row * slope + icpt - 1.0) / 10.0)
    return out


def nesz_flattening(noise, inc):
    """Flatten a (line, sample) noise-equivalent sigma0 by a per-line degree-1 fit of its dB value
    against incidence; NaNs are first replaced by the column mean.  Returns 10**((fit - 1)/10), float64.

    >>> nesz_flat = nesz_flattening(nesz_cr, inc)
    >>> dsig_cr = (1.25 / (sigma0_cr / nesz_flat)) ** 4.0
    """
    if noise.ndim != 2:
        raise IndexError("Only 2D noise allowed")
    from .. import _device, _lib, options
    if _device.any_device_array(noise, inc):  # rasters resident in HBM: a float64 torch tensor comes back, asynchronously
        import torch
        dev = _device.device_of(noise, inc)
        t_n, t_i = _device.as_tensor(noise, dev), _device.as_tensor(inc, dev)
        dt = torch.float32 if (t_n.dtype == torch.float32 and t_i.dtype == torch.float32) else torch.float64
        t_n, t_i = t_n.to(dt).contiguous(), t_i.to(dt).expand(t_n.shape).contiguous()
        out = torch.empty(t_n.shape, dtype=torch.float64, device=dev)
        if t_n.numel():
            ctx = _lib.default_context(dev.index if dev.index is not None else torch.cuda.current_device())
            with _device.on_current_stream(ctx, dev):
                ctx.nesz_flatten_raw(t_n.shape[0], t_n.shape[1], _device.xsw_dtype(t_n), _lib.MEM_DEVICE, t_n.data_ptr(), t_i.data_ptr(),
                                     out.data_ptr())
                for t in (t_n, t_i):
                    t.record_stream(torch.cuda.current_stream(dev))
        return out
    values, inc_v = np.asarray(noise), np.asarray(inc)
    mode = options.nesz_on_device
    # "auto" is parity-first like the other defaults: float64 rasters only (device == host route to ~1e-13); float32 rasters
    # -- where the reference itself accumulates in float32 and the float64 device sums differ from it by ~1e-6 -- go to the
    # device on request only ("device")
    on_dev = mode == "device" or (mode == "auto" and values.size >= options.nesz_device_min_size
                                  and values.dtype == np.float64 and inc_v.dtype == np.float64
                                  and _lib.device_count_safe() > 0)
    if on_dev and values.dtype in (np.float32, np.float64) and inc_v.shape == values.shape and values.size:
        return _lib.default_context(options.device).nesz_flatten_host(values, inc_v)
    return _nesz_flattening_host(values, inc_v)
