"""Cross-pol preprocessing helpers used right before `invert_from_model`
(reference: src/xsarsea/windspeed/utils.py): `get_dsig` (:47-91), `get_dsig_wspd` (:18-44),
`nesz_flattening` (:94-163).  Elementwise formulas / one degree-1 fit per row: host numpy."""
import warnings

import numpy as np

_DSIG_WSPD = {
    "dsig_wspd_rs2_v3": (-0.4908643753212401, 16.763199934792965, 1.3891445172991084, 20.616914824394343),
    "dsig_wspd_s1_ew_rec_v3": (-0.5858970325653666, 16.50039320910609, 1.1032031322520397, 7.434663633997121),
    "dsig_wspd_rcm_v3": (-0.7920301376936547, 15.8288289109038, 0.24040294696606557, 0.2538177092195224),
}


def get_dsig_wspd(name, U_crosspol, SNR_cr):
    """Weight alpha(U, SNR) in [0, 1]: logistic in (U - c0 + gamma*SNR) times a roll-off above Umax = 30."""
    b, c0_base, gamma, k = _DSIG_WSPD[name]
    core = 1 / (1 + np.exp(-b * (U_crosspol - (c0_base - gamma * SNR_cr))))
    drop = 1 / (1 + np.exp((U_crosspol - 30) * k))
    return np.clip(core * drop, 0, 1)


def get_dsig(name, inc, sigma0_cr, nesz_cr):
    """`dsig_cr` for `invert_from_model` from the cross-pol signal-to-noise ratio."""
    snr = sigma0_cr / nesz_cr
    if name == "gmf_s1_v2":
        c = 1.46852088 + 1.4058646 / (1 + np.exp(-1.57952257 * (inc - 25.61843791)))
        return 1 / np.sqrt(1 * snr ** c)
    if name == "gmf_rs2_v2":
        return 1 / np.sqrt(1 * snr ** 8)
    if name in ("sarwing_lut_cmodms1ahw", "nc_lut_cmodms1ahw"):
        return (1.25 / snr) ** 4.0
    raise ValueError("dsig names different than 'gmf_s1_v2' or 'gmf_rs2_v2' or 'sarwing_lut_cmodms1ahw' or "
                     "'nc_lut_cmodms1ahw' are not handled. You can compute your own dsig_cr.")


def nesz_flattening(noise, inc):
    """Flatten a (line, sample) noise-equivalent sigma0 by a per-line degree-1 fit of its dB value
    against incidence; NaNs are first replaced by the column mean.  Returns 10**((fit - 1)/10)."""
    if noise.ndim != 2:
        raise IndexError("Only 2D noise allowed")
    values = np.asarray(noise, dtype=np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        col_mean = np.nanmean(values, axis=0)
        inc_row = np.nanmean(np.asarray(inc, dtype=np.float64), axis=0)
    out = np.empty_like(values)
    for i, row in enumerate(values):
        filled = np.where(np.isnan(row), col_mean, row)
        with np.errstate(all="ignore"):
            db = 10.0 * np.log10(filled)
        ok = np.isfinite(db)
        try:
            slope, icpt = np.polyfit(inc_row[ok], db[ok], 1)
        except TypeError:
            out[i] = np.nan
            continue
        out[i] = 10.0 ** ((inc_row * slope + icpt - 1.0) / 10.0)
    return out
