"""Oracle restatement of the cross-pol preprocessing helpers (reference: src/xsarsea/windspeed/utils.py),
numpy inputs.  TEST INFRASTRUCTURE ONLY.

Pinned bit for bit by tests/golden/crosspol_prep.npz, produced by executing the reference's own
`get_dsig` / `get_dsig_wspd` / `nesz_flattening` (tests/golden/make_golden.py crosspol_prep).
"""
import warnings

import numpy as np


def get_dsig_wspd(name, U_crosspol, SNR_cr):
    """utils.py:18-44: alpha = clip(logistic(U - (c0 - gamma SNR); b) * rolloff(U; Umax = 30, k), 0, 1)."""
    b, c0_base, gamma, k = {  # :27-43
        "dsig_wspd_rs2_v3": (-0.4908643753212401, 16.763199934792965, 1.3891445172991084, 20.616914824394343),
        "dsig_wspd_s1_ew_rec_v3": (-0.5858970325653666, 16.50039320910609, 1.1032031322520397, 7.434663633997121),
        "dsig_wspd_rcm_v3": (-0.7920301376936547, 15.8288289109038, 0.24040294696606557, 0.2538177092195224),
    }[name]
    c0 = c0_base - gamma * SNR_cr                        # :20
    alpha_core = 1 / (1 + np.exp(-b * (U_crosspol - c0)))  # :21-22
    drop = 1 / (1 + np.exp((U_crosspol - 30) * k))       # :23
    return np.clip(alpha_core * drop, 0, 1)              # :24


def get_dsig(name, inc, sigma0_cr, nesz_cr):
    """utils.py:47-91."""
    if name == "gmf_s1_v2":  # :66-75: c = sigmoid(inc; float64 coefficient array), 1/sqrt(snr**c)
        c0, c1, d0, d1 = np.array([1.57952257, 25.61843791, 1.46852088, 1.4058646])
        c = d0 + d1 / (1 + np.exp(-c0 * (inc - c1)))
        return 1 / np.sqrt(1 * (sigma0_cr / nesz_cr) ** c)
    if name == "gmf_rs2_v2":  # :77-80
        return 1 / np.sqrt(1 * (sigma0_cr / nesz_cr) ** 8)
    if name in ("sarwing_lut_cmodms1ahw", "nc_lut_cmodms1ahw"):  # :82-86
        return (1.25 / (sigma0_cr / nesz_cr)) ** 4.0
    raise ValueError(name)  # :88-91


def nesz_flattening(noise, inc):
    """utils.py:94-163: column nan-mean fill, per-line degree-1 polyfit of the dB noise against the mean incidence row."""
    if noise.ndim != 2:  # :116-117
        raise IndexError("Only 2D noise allowed")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        noise_mean = np.nanmean(noise, axis=0)  # :119-122

        def one_row(noise_row, inc_row):  # :131-155
            flat = noise_row.copy()
            flat[np.isnan(flat)] = noise_mean[np.isnan(flat)]
            db = 10.0 * np.log10(flat)
            try:
                coef = np.polyfit(inc_row[np.isfinite(db)], db[np.isfinite(db)], 1)
            except TypeError:
                return np.full(noise_row.shape, np.nan)
            return 10.0 ** ((inc_row * coef[0] + coef[1] - 1.0) / 10.0)

        return np.apply_along_axis(one_row, 1, noise, np.nanmean(inc, axis=0))  # :163
