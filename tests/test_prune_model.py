"""CPU: the search-window logic the device kernel follows (tests/prune_model.py is its executable
specification) never excludes the oracle's argmin -- property test on several direction axes."""
import numpy as np
import pytest

import prune_model as pm
from oracle import cport, gmf
from oracle import invert as oinv
from oracle import lut as olut


@pytest.mark.parametrize("phimax,nphi", [(180, 181), (360, 361), (90, 91), (170, 86)])
def test_window_contains_argmin(phimax, nphi):
    rng = np.random.default_rng(nphi)
    inc_ax, w_ax, phi_ax = np.linspace(20, 44, 9), np.linspace(0.5, 39.5, 118), np.linspace(0, phimax, nphi)
    co = 10 * np.log10(gmf.gmf_cmod5n(inc_ax[:, None, None], w_ax[None, :, None], phi_ax[None, None, :]) + 1e-15)
    co = co + 0.05 * rng.standard_normal(co.shape)
    lco = olut.Lut(co, inc_ax, w_ax, phi_ax, "dB", "x", "co", "VV")
    p = oinv.Prepared(lco, None)
    n = 1500
    inc, wt, pt = rng.uniform(18, 46, n), rng.uniform(0.5, 35, n), rng.uniform(-180, 180, n)
    s = oinv.to_db(gmf.gmf_cmod5n(inc, wt, pt) * rng.gamma(100, 1 / 100, n))
    anc = wt * np.exp(1j * np.deg2rad(pt)) + rng.normal(0, 1.5, n) + 1j * rng.normal(0, 1.5, n)
    anc[:200] = rng.uniform(0, 40, 200) * np.exp(1j * rng.uniform(-np.pi, np.pi, 200))  # ancillary far off
    anc[200:220] *= 1e-4
    nan = np.full(n, np.nan)
    idx = cport.invert_numpy(p, inc, s, nan, nan, anc, return_idx=True, reference_layout=False)[2]
    cphi, sphi = np.cos(np.radians(phi_ax)), np.sin(np.radians(phi_ax))
    evaluated = []
    for i in range(n):
        ii = np.argmin(np.abs(inc_ax - inc[i]))
        r = pm.pruned_argmin(co[ii], w_ax, phi_ax, cphi, sphi, p.phi_180, s[i], anc[i].real, anc[i].imag, 0.1)
        assert (r[0], r[1]) == (idx[i, 0], idx[i, 1]), (i, r, idx[i])
        evaluated.append(r[2])
    assert np.mean(evaluated) < 0.7 * len(w_ax) * nphi
