"""Oracle restatement of the reference's wind inversion (test infrastructure).

Restates `/root/reference/src/xsarsea/windspeed/windspeed.py`:
  * `to_db` ................... :126-130   10*log10(sigma0 + 1e-15), dtype follows the input
  * `prepare_luts` ............ :144-181   LUT transposes, phi_180 detection, candidate vectors
  * `invert_1d` ............... :183-282   THE per-pixel kernel (`__invert_from_model_1d`), same
                                           dense float64 temporaries, same operation order, numpy
                                           `argmin` first-minimum tie rule
  * `invert_numpy` ............ :132-331   gufunc contract (:306-323): float64/complex128 in,
                                           two complex128 out, any leading shape
  * `invert_from_model` ....... :18-131, :390-439  argument routing for plain numpy inputs
                                           (mono co-pol / mono cross-pol / dual-pol), final dual-pol
                                           select (:426-428)

LUT objects are `oracle.lut.Lut` (dB units, values shaped (incidence, wspd[, phi])).
"""
import warnings

import numpy as np


def to_db(sigma0):
    """windspeed.py:126-130 (numpy branch; the dtype of the result follows the input's)."""
    with np.errstate(all="ignore"):
        return 10 * np.log10(sigma0 + 1e-15)


class Prepared:
    """Closure state of `_invert_from_model_numpy` (windspeed.py:139-181)."""

    def __init__(self, lut_co, lut_cr, dsig_co=0.1):
        self.dsig_co = dsig_co
        self.d_antenna = 2
        self.d_azi = 2
        self.dwspd_fg = 2
        if lut_co is not None:
            # (wspd, phi, incidence) C-contiguous, windspeed.py:145-147
            self.co_lut = np.ascontiguousarray(np.transpose(lut_co.values, (1, 2, 0)))
            self.wspd_dim = np.asarray(lut_co.wspd, dtype=np.float64)
            self.phi_dim = np.asarray(lut_co.phi, dtype=np.float64)
            self.inc_dim = np.asarray(lut_co.incidence, dtype=np.float64)
            self.phi_180 = bool((180 - (self.phi_dim[-1] - self.phi_dim[0])) < 2)  # :152-156
        else:  # :157-164
            self.co_lut = np.array([[[]]], dtype=np.float64)
            self.wspd_dim = np.array([], dtype=np.float64)
            self.phi_dim = np.array([], dtype=np.float64)
            self.inc_dim = np.array([], dtype=np.float64)
            self.phi_180 = False
        self.phi_lut, self.wspd_lut = np.meshgrid(self.phi_dim, self.wspd_dim)  # (wspd, phi) :166
        self.lut_co_antenna = self.wspd_lut * np.cos(np.radians(self.phi_lut))  # :167
        self.lut_co_azi = self.wspd_lut * np.sin(np.radians(self.phi_lut))  # :168
        if lut_cr is not None:  # :170-176
            self.cr_lut = np.ascontiguousarray(np.transpose(lut_cr.values, (1, 0)))
            self.wspd_cr = np.asarray(lut_cr.wspd, dtype=np.float64)
            self.inc_cr_dim = np.asarray(lut_cr.incidence, dtype=np.float64)
        else:
            self.cr_lut = np.array([[]], dtype=np.float64)
            self.wspd_cr = np.array([], dtype=np.float64)
            self.inc_cr_dim = np.array([], dtype=np.float64)


def invert_1d(p, inc_1d, sigma0_co_db_1d, sigma0_cr_db_1d, dsig_cr_1d, ancillary_wind_1d,
              out_co, out_cr, idx_out=None):
    """windspeed.py:183-282.  `idx_out` (optional int64[n, 3]) receives (i_wspd, i_phi, i_wspd_cr),
    -1 where no search ran: an oracle-side convenience for index-exact parity checks."""
    for i in range(len(inc_1d)):
        one_inc = inc_1d[i]
        one_sigma0_co_db = sigma0_co_db_1d[i]
        one_sigma0_cr_db = sigma0_cr_db_1d[i]
        one_dsig_cr = dsig_cr_1d[i]
        one_ancillary_wind = ancillary_wind_1d[i]
        if idx_out is not None:
            idx_out[i, :] = -1

        if np.isnan(one_inc):  # :198-201
            out_co[i] = np.nan
            out_cr[i] = np.nan
            continue

        if not np.isnan(np.abs(one_sigma0_co_db)) and np.isnan(np.abs(one_ancillary_wind)):  # :204-207
            out_co[i] = np.nan
            out_cr[i] = np.nan
            continue

        if not np.isnan(one_sigma0_co_db):  # :209-247
            i_inc = np.argmin(np.abs(p.inc_dim - one_inc))
            lut_inc = p.co_lut[:, :, i_inc]
            m_antenna = np.real(one_ancillary_wind)
            m_azi = np.imag(one_ancillary_wind)
            if p.phi_180:
                m_azi = np.abs(m_azi)
            Jwind_co = ((p.lut_co_antenna - m_antenna) / p.d_antenna) ** 2 + (
                (p.lut_co_azi - m_azi) / p.d_azi
            ) ** 2
            Jsig_co = ((lut_inc - one_sigma0_co_db) / p.dsig_co) ** 2
            J_co = Jwind_co + Jsig_co
            iJ_co = np.argmin(J_co)
            lut_idx = (iJ_co // J_co.shape[-1], iJ_co % J_co.shape[-1])
            if idx_out is not None:
                idx_out[i, 0], idx_out[i, 1] = lut_idx
            wspd_co = p.wspd_lut[lut_idx]
            wphi_co = p.phi_lut[lut_idx]
            if p.phi_180:  # :234-242
                sol = wspd_co * np.exp(1j * np.deg2rad(wphi_co))
                sol_2 = wspd_co * np.exp(1j * (np.deg2rad(-wphi_co)))
                diff_angle = np.angle(one_ancillary_wind / sol)
                diff_angle_2 = np.angle(one_ancillary_wind / sol_2)
                wind_co = sol if np.abs(diff_angle) <= np.abs(diff_angle_2) else sol_2
            else:
                wind_co = wspd_co * np.exp(1j * np.deg2rad(wphi_co))
        else:
            wind_co = np.nan * 1j  # :250

        if not np.isnan(one_sigma0_cr_db) and not np.isnan(one_dsig_cr):  # :252-276
            i_inc = np.argmin(np.abs(p.inc_cr_dim - one_inc))
            lut_cr_inc = p.cr_lut[:, i_inc]
            Jwind_cr = ((p.wspd_cr - np.abs(wind_co)) / p.dwspd_fg) ** 2.0
            Jsig_cr = ((lut_cr_inc - one_sigma0_cr_db) / one_dsig_cr) ** 2.0
            if not np.isnan(np.abs(wind_co)):
                J_cr = Jsig_cr + Jwind_cr
            else:
                J_cr = Jsig_cr
            i_cr = np.argmin(J_cr)
            if idx_out is not None:
                idx_out[i, 2] = i_cr
            wspd_dual = p.wspd_cr[i_cr]
            if not np.isnan(np.abs(wind_co)):
                phi_dual = np.angle(wind_co)
            else:
                phi_dual = 0
            wind_dual = wspd_dual * np.exp(1j * phi_dual)
        else:
            wind_dual = np.nan * 1j  # :278

        out_co[i] = wind_co
        out_cr[i] = wind_dual


def invert_numpy(p, np_inc, np_sigma0_co_db, np_sigma0_cr_db, np_dsig_cr, np_ancillary_wind,
                 return_idx=False):
    """gufunc contract of windspeed.py:306-323: inputs cast to float64/complex128, flattened over
    all axes (the core dimension is the last axis; pixels are independent), two complex128 outs."""
    shape = np.shape(np_inc)
    args = [
        np.ascontiguousarray(np.broadcast_to(np.asarray(a), shape)).astype(t).ravel()
        for a, t in zip(
            (np_inc, np_sigma0_co_db, np_sigma0_cr_db, np_dsig_cr, np_ancillary_wind),
            (np.float64, np.float64, np.float64, np.float64, np.complex128),
        )
    ]
    n = args[0].size
    out_co = np.empty(n, dtype=np.complex128)
    out_cr = np.empty(n, dtype=np.complex128)
    idx = np.empty((n, 3), dtype=np.int64) if return_idx else None
    with warnings.catch_warnings(), np.errstate(all="ignore"):
        warnings.simplefilter("ignore", RuntimeWarning)
        invert_1d(p, *args, out_co, out_cr, idx)
    res = (out_co.reshape(shape), out_cr.reshape(shape))
    if return_idx:
        res = res + (idx.reshape(shape + (3,)),)
    return res


def invert_from_model(inc, sigma0, sigma0_dual=None, /, ancillary_wind=None, dsig_co=0.1, dsig_cr=0.1,
                      lut_co=None, lut_cr=None, return_idx=False):
    """windspeed.py:18-131 + :390-439 for plain numpy inputs.

    `lut_co` / `lut_cr` are dB `oracle.lut.Lut` objects (the reference resolves them from `model=`
    through `get_model(...).to_lut(units="dB", **kwargs)`, :78-83, :144, :171).
    Mono co-pol: pass lut_co only.  Mono cross-pol: pass lut_cr only.  Dual: both + sigma0_dual.
    """
    nan = sigma0 * np.nan  # :71
    if ancillary_wind is None:
        ancillary_wind = nan
    if sigma0_dual is None:
        if lut_co is not None:  # co-pol model (:103-107)
            sigma0_co, sigma0_cr = sigma0, nan
            assert np.any(~np.isnan(ancillary_wind))
        else:  # cross-pol model (:108-117)
            sigma0_co, sigma0_cr = nan, sigma0
    else:
        sigma0_co, sigma0_cr = sigma0, sigma0_dual
    if np.isscalar(dsig_cr):
        dsig_cr = sigma0_cr * 0 + dsig_cr  # :122-123
    sigma0_co_db = to_db(sigma0_co)
    sigma0_cr_db = to_db(sigma0_cr) if sigma0_cr is not nan else nan
    use_cr = not np.all(np.isnan(sigma0_cr_db))  # :170
    p = Prepared(lut_co, lut_cr if use_cr else None, dsig_co)
    res = invert_numpy(p, inc, sigma0_co_db, sigma0_cr_db, dsig_cr, ancillary_wind, return_idx)
    ws_co, ws_cr_or_dual = res[0], res[1]
    if sigma0_dual is None:
        out = ws_co if lut_co is not None else np.abs(ws_cr_or_dual)  # :415-423
    else:
        with np.errstate(all="ignore"):
            wspd_dual = np.where((np.abs(ws_co) < 5) | (np.abs(ws_cr_or_dual) < 5), ws_co, ws_cr_or_dual)
        out = (ws_co, wspd_dual)  # :426-439
    if return_idx:
        return out, res[2]
    return out
