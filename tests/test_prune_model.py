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


@pytest.mark.parametrize("seed", range(4))
def test_search_window_contains_disc(seed):
    """Every grid candidate inside the disc |c - m| <= 2 sqrt(j_ub) lies inside the index box -- also when
    the disc boundary passes exactly through grid points (mag, R multiples of the steps)."""
    rng = np.random.default_rng(seed)
    for case in range(300):
        n_w, n_phi = int(rng.integers(2, 60)), int(rng.integers(2, 90))
        w0, wstep = rng.choice([0.2, 0.5, 3.0]), rng.choice([0.1, 0.25, 1.0])
        phi0 = rng.choice([0.0, 0.0, -180.0, 10.0])
        span = rng.choice([180.0, 360.0, 90.0, 170.0, 359.0])
        w_ax = w0 + wstep * np.arange(n_w)
        phi_ax = np.linspace(phi0, phi0 + span, n_phi)
        inv_wstep, inv_dphi = (n_w - 1) / (w_ax[-1] - w_ax[0]), (n_phi - 1) / (phi_ax[-1] - phi_ax[0])
        if case % 3 == 0:  # boundary through grid points
            mag = w_ax[rng.integers(0, n_w)]
            R = wstep * rng.integers(0, 12)
            theta = phi_ax[rng.integers(0, n_phi)]
        else:
            mag, R, theta = rng.uniform(0, 1.3 * w_ax[-1]), rng.uniform(0, 8) ** 2 / 8, rng.uniform(phi0, phi0 + 360)
        if theta < phi0:
            theta += 360.0
        j_ub = (R / 2) ** 2
        a, b = mag * np.cos(np.radians(theta)), mag * np.sin(np.radians(theta))
        w_lo, w_hi, ip_lo, ip_hi = pm.search_window(mag, theta, j_ub, w0, inv_wstep, n_w, phi0, phi_ax[-1], inv_dphi, n_phi)
        cx = w_ax[:, None] * np.cos(np.radians(phi_ax))[None, :]
        cy = w_ax[:, None] * np.sin(np.radians(phi_ax))[None, :]
        inside = ((cx - a) ** 2 + (cy - b) ** 2) / 4 <= j_ub
        iw, ip = np.nonzero(inside)
        if iw.size:
            assert w_lo <= iw.min() and iw.max() <= w_hi, (case, mag, R, theta, w_lo, w_hi, iw.min(), iw.max())
            assert ip_lo <= ip.min() and ip.max() <= ip_hi, (case, mag, R, theta, ip_lo, ip_hi, ip.min(), ip.max())


@pytest.mark.parametrize("kind", ["smooth", "plateaus", "rolloff", "noisy"])
def test_band_rule_never_excludes_the_argmin(kind):
    """Round-2 band pruning (|L - s| <= dsig sqrt(J_ub), one row interval per direction on monotone columns): same argmin
    as the oracle on a smooth LUT, a LUT quantised to 0.05 dB (long exact plateaus), a LUT that rolls off at high wind
    (the window must stay inside the monotone rows or fall back) and a noisy LUT (eligible only on its steep low-wind rows: nearly always falls back)."""
    rng = np.random.default_rng({"smooth": 1, "plateaus": 2, "rolloff": 3, "noisy": 4}[kind])
    inc_ax, w_ax, phi_ax = np.linspace(20, 44, 7), np.linspace(0.5, 39.5, 196), np.linspace(0, 180, 91)
    co = 10 * np.log10(gmf.gmf_cmod5n(inc_ax[:, None, None], w_ax[None, :, None], phi_ax[None, None, :]) + 1e-15)
    if kind == "plateaus":
        co = np.round(co / 0.05) * 0.05
    elif kind == "rolloff":
        co = co - 0.02 * np.maximum(w_ax[None, :, None] - 24.0, 0.0) ** 2
    elif kind == "noisy":
        co = co + 0.05 * rng.standard_normal(co.shape)
    lco = olut.Lut(co, inc_ax, w_ax, phi_ax, "dB", "x", "co", "VV")
    p = oinv.Prepared(lco, None)
    n = 1200
    inc, wt, pt = rng.uniform(18, 46, n), rng.uniform(0.5, 35, n), rng.uniform(-180, 180, n)
    s = oinv.to_db(gmf.gmf_cmod5n(inc, wt, pt) * rng.gamma(100, 1 / 100, n))
    anc = wt * np.exp(1j * np.deg2rad(pt)) + rng.normal(0, 1.5, n) + 1j * rng.normal(0, 1.5, n)
    anc[:150] = rng.uniform(0, 40, 150) * np.exp(1j * rng.uniform(-np.pi, np.pi, 150))
    nan = np.full(n, np.nan)
    idx = cport.invert_numpy(p, inc, s, nan, nan, anc, return_idx=True, reference_layout=False)[2]
    cphi, sphi = np.cos(np.radians(phi_ax)), np.sin(np.radians(phi_ax))
    used, evaluated = 0, []
    for i in range(n):
        ii = np.argmin(np.abs(inc_ax - inc[i]))
        r = pm.band_pruned_argmin(co[ii], w_ax, phi_ax, cphi, sphi, p.phi_180, s[i], anc[i].real, anc[i].imag, 0.1)
        assert (r[0], r[1]) == (idx[i, 0], idx[i, 1]), (kind, i, r, idx[i])
        used += r[3]
        if r[3]:
            evaluated.append(r[2])
    if kind == "noisy":
        assert used < 0.1 * n  # only the steep low-wind rows of a noisy LUT are monotone
    else:
        assert used > 0.3 * n and np.mean(evaluated) < 120, (used, np.mean(evaluated))
    # CMOD5.N itself saturates and decreases at high wind / low incidence: the monotone prefix is a per-slice property
    assert pm.mono_rows(co[0]) < len(w_ax) and (kind in ("noisy", "rolloff") or pm.mono_rows(co[-1]) == len(w_ax))


@pytest.mark.parametrize("kind", ["cmod5n", "rolloff"])
def test_tail_cut_keeps_saturating_windows_with_the_band_rule(kind):
    """Round 3: an a-priori wind well above the one sigma0 points to puts the window's upper rows on the saturated top of
    CMOD5.N (past the monotone rows).  When every LUT value up there lies above s + d the window is cut at the last monotone row
    (L.tail_min): same argmin as the oracle, and clearly more pixels decided by the band rule than without the cut.  A top that
    rolls off steeply falls back below the observed sigma0: no cut there."""
    rng = np.random.default_rng({"cmod5n": 11, "rolloff": 12}[kind])
    inc_ax, w_ax, phi_ax = np.linspace(18, 30, 7), np.linspace(0.5, 79.5, 396), np.linspace(0, 180, 91)
    co = 10 * np.log10(gmf.gmf_cmod5n(inc_ax[:, None, None], w_ax[None, :, None], phi_ax[None, None, :]) + 1e-15)
    if kind == "rolloff":
        co = co - 0.004 * np.maximum(w_ax[None, :, None] - 30.0, 0.0) ** 2
    assert all(pm.mono_rows(co[k]) < len(w_ax) for k in range(len(inc_ax)))
    lco = olut.Lut(co, inc_ax, w_ax, phi_ax, "dB", "x", "co", "VV")
    p = oinv.Prepared(lco, None)
    n = 900
    inc, wt, pt = rng.uniform(18, 30, n), rng.uniform(3, 22, n), rng.uniform(-180, 180, n)
    s = oinv.to_db(gmf.gmf_cmod5n(inc, wt, pt) * rng.gamma(100, 1 / 100, n))
    anc = rng.uniform(1.2, 2.6, n) * wt * np.exp(1j * np.deg2rad(pt + rng.normal(0, 10, n)))  # a-priori speed 1.2 .. 2.6 x the truth
    nan = np.full(n, np.nan)
    idx = cport.invert_numpy(p, inc, s, nan, nan, anc, return_idx=True, reference_layout=False)[2]
    cphi, sphi = np.cos(np.radians(phi_ax)), np.sin(np.radians(phi_ax))
    used_cut = used_plain = used_whole = 0
    for i in range(n):
        ii = np.argmin(np.abs(inc_ax - inc[i]))
        r = pm.band_pruned_argmin(co[ii], w_ax, phi_ax, cphi, sphi, p.phi_180, s[i], anc[i].real, anc[i].imag, 0.1, max_len=64)
        assert (r[0], r[1]) == (idx[i, 0], idx[i, 1]), (kind, i, r, idx[i])
        used_cut += r[3]
        used_plain += pm.band_pruned_argmin(co[ii], w_ax, phi_ax, cphi, sphi, p.phi_180, s[i], anc[i].real, anc[i].imag, 0.1,
                                            max_len=64, tail_cut=False)[3]
        used_whole += pm.band_pruned_argmin(co[ii], w_ax, phi_ax, cphi, sphi, p.phi_180, s[i], anc[i].real, anc[i].imag, 0.1,
                                            max_len=64, tail_cut="whole")[3]
    # tail sweep (k_invert_band2): windows the cut cannot take keep the band rule on their monotone part, the rows past it are all
    # candidates: same argmin, and more pixels stay with the band rule still
    used_sweep = 0
    for i in range(n):
        ii = np.argmin(np.abs(inc_ax - inc[i]))
        r = pm.band_pruned_argmin(co[ii], w_ax, phi_ax, cphi, sphi, p.phi_180, s[i], anc[i].real, anc[i].imag, 0.1, max_len=64, tail_sweep=256)
        assert (r[0], r[1]) == (idx[i, 0], idx[i, 1]), (kind, "tail sweep", i, r, idx[i])
        used_sweep += r[3]
    assert used_sweep > used_cut, (used_sweep, used_cut)
    if kind == "cmod5n":  # and the minimum over the window's own directions cuts more windows than the one over all directions
        assert used_cut > used_whole > used_plain + 0.05 * n, (used_cut, used_whole, used_plain)
    else:  # the rolled-off top falls back below every observed sigma0: the cut must never fire there
        assert used_cut == used_whole == used_plain, (used_cut, used_whole, used_plain)


@pytest.mark.parametrize("kind", ["smooth", "quantised", "steps", "flat"])
def test_inverse_row_table_interval_is_a_tight_superset_of_the_band(kind):
    """The rows a lane reads off the inverse-row table always contain the exact band  {r in window: s - d <= col[r] <= s + d}
    of a monotone column -- for thresholds ON table values, ON grid thresholds, below / above the column, and windows
    clipped anywhere -- and exceed it by no more than the rows that share a grid bin with its ends (thresholds inside the grid)."""
    import prune_model as pm
    rng = np.random.default_rng({"smooth": 1, "quantised": 2, "steps": 3, "flat": 4}[kind])
    bins = 256
    for case in range(300):
        n = int(rng.integers(2, 200))
        if kind == "flat":
            col = np.full(n, rng.normal())
        else:
            col = np.cumsum(rng.gamma(0.7, 0.08, n)) + rng.normal(-20, 5)
            if kind == "quantised":
                col = np.round(col / 0.25) * 0.25
            if kind == "steps":
                col = np.repeat(col[:: 7], 7)[:n] if n >= 7 else col
        mono = n if rng.random() < 0.7 else int(rng.integers(1, n + 1))
        lo, hi = col[:mono].min(), col[:mono].max()
        width = (hi - lo) / bins
        ok = width > 0
        t0, width, inv_width = (lo, width, 1.0 / width) if ok else (0.0, 0.0, 0.0)
        inv = pm.inverse_rows(col, mono, t0, width, bins)
        assert inv[0] == 0 and np.all(np.diff(inv) >= 0) and inv[-1] <= mono
        for _ in range(20):
            mode = rng.integers(0, 5)
            if mode == 0:
                s = col[rng.integers(0, mono)]
            elif mode == 1:
                s = t0 + rng.integers(0, bins + 1) * width
            elif mode == 2:
                s = rng.uniform(lo - 1, hi + 1)
            elif mode == 3:
                s = lo - rng.uniform(0, 3)
            else:
                s = hi + rng.uniform(0, 3)
            d = float(rng.choice([0.0, 1e-12, 0.01, 0.3, 5.0]))
            thr_lo, thr_hi = s - d, s + d
            w_hi = int(rng.integers(0, mono))            # eligible windows end inside the monotone rows
            w_lo = int(rng.integers(0, w_hi + 1))
            b_lo, b_hi = pm.table_bins(t0, width, inv_width, bins, thr_lo, thr_hi)
            assert b_lo == 0 or b_lo * width + t0 <= thr_lo
            assert b_hi == -1 or b_hi * width + t0 > thr_hi
            first, last = pm.band_rows_from_table(inv, b_lo, b_hi, w_lo, w_hi)
            rows = np.arange(w_lo, w_hi + 1)
            exact = rows[(col[rows] >= thr_lo) & (col[rows] <= thr_hi)]
            if exact.size:
                assert first <= exact[0] and exact[-1] <= last, (kind, case, s, d)
            # tightness: rows read but outside the band lie within one grid bin of its ends (or the grid is degenerate)
            if ok and last >= first:
                extra = [r for r in range(first, last + 1) if not (thr_lo <= col[r] <= thr_hi)]
                for r in extra:
                    assert (col[r] < thr_lo and col[r] >= thr_lo - width * (1 + 1e-9) - 1e-12 and b_lo > 0) or \
                           (col[r] < thr_lo and b_lo in (0, bins - 1)) or \
                           (col[r] > thr_hi and (b_hi == -1 or col[r] < thr_hi + width * (1 + 1e-9) + 1e-12)), (kind, case, r)


@pytest.mark.parametrize("seed", range(4))
def test_chord_rows_contain_the_disc(seed):
    """k_invert_band2's chord clip: along every direction, every grid speed whose wind term alone is <= j_ub lies inside the
    clipped row interval (and a direction reported as missing the disc has none) -- also with the disc's boundary exactly on
    grid points, tiny and huge discs, m at the origin, and J_ub recovered from the rounded band thresholds as on the device."""
    rng = np.random.default_rng(100 + seed)
    for case in range(400):
        n_w, n_phi = int(rng.integers(2, 120)), int(rng.integers(2, 90))
        w0, wstep = rng.choice([0.2, 0.5, 3.0]), rng.choice([0.1, 0.25, 1.0])
        w_ax = w0 + wstep * np.arange(n_w)
        phi_ax = np.linspace(0.0, rng.choice([180.0, 360.0]), n_phi)
        inv_wstep = (n_w - 1) / (w_ax[-1] - w_ax[0])
        if case % 3 == 0:  # boundary through grid points
            mag, R, theta = w_ax[rng.integers(0, n_w)], wstep * rng.integers(0, 12), phi_ax[rng.integers(0, n_phi)]
        elif case % 7 == 1:
            mag, R, theta = 0.0, rng.uniform(0, 30), 0.0
        else:
            mag, R, theta = rng.uniform(0, 1.3 * w_ax[-1]), rng.uniform(0, 8) ** 2 / 8 * rng.choice([1e-3, 1.0, 1.0, 10.0]), rng.uniform(0, 360)
        j_true = (R / 2) ** 2
        a, b = mag * np.cos(np.radians(theta)), mag * np.sin(np.radians(theta))
        dsig = float(rng.choice([0.01, 0.1, 1.0, 5.0]))
        s = rng.uniform(-40.0, 5.0)
        j_ub = j_true * (1.0 + 1e-9) + 1e-9               # co_window_lanes
        d = pm.band_radius(j_ub, dsig)
        thr_lo, thr_hi = s - d, s + d                      # what the slot holds (rounded)
        for ip in range(n_phi):
            c, sn = np.cos(np.radians(phi_ax[ip])), np.sin(np.radians(phi_ax[ip]))
            inside = np.nonzero(((w_ax * c - a) ** 2 + (w_ax * sn - b) ** 2) / 4 <= j_true)[0]
            rows = pm.chord_rows(0.5 * a, 0.5 * b, c, sn, thr_lo, thr_hi, 1.0 / dsig, w0, inv_wstep)
            if rows is None:
                assert inside.size == 0, (case, ip, mag, R, theta)
            else:
                if inside.size:
                    assert rows[0] <= inside.min() and inside.max() <= rows[1], (case, ip, mag, R, theta, rows, inside.min(), inside.max())
                # ... and it is tight: at most two rows of slack at either end (the inflations are ~1e-6 of the radius)
                kept = max(min(rows[1], n_w - 1) - max(rows[0], 0) + 1, 0)
                assert kept <= inside.size + 4, (case, ip, mag, R, theta, rows, inside.size)


# ----------------------------------------------------------------------------------------------------------------------
# Round 4: the block pyramid (co_block_search of the general kernel; tests/prune_model.py: block_pruned_argmin)
@pytest.mark.parametrize("seed", range(4))
def test_block_cell_bound_is_a_lower_bound(seed):
    """cell_wind_lb <= the smallest |c - m|^2 / 4 over the grid points of a block, whatever the axis (0..180, 0..360, -180..180,
    coarse steps, spans beyond 170 deg fall back to the radial bound), also with m on a block edge, on a grid point, at the origin."""
    rng = np.random.default_rng(100 + seed)
    for case in range(400):
        n_w, n_phi = int(rng.integers(2, 70)), int(rng.integers(2, 120))
        w0, wstep = rng.choice([0.2, 0.5, 3.0]), rng.choice([0.1, 0.25, 1.0])
        phi0 = rng.choice([0.0, 0.0, -180.0, 10.0])
        span = rng.choice([180.0, 360.0, 90.0, 170.0, 359.0])
        w_ax, phi_ax = w0 + wstep * np.arange(n_w), np.linspace(phi0, phi0 + span, n_phi)
        cphi, sphi = np.cos(np.radians(phi_ax)), np.sin(np.radians(phi_ax))
        dphi = span / (n_phi - 1)
        kind = case % 4
        if kind == 0:  # m on a grid point
            mag, th = w_ax[rng.integers(0, n_w)], np.radians(phi_ax[rng.integers(0, n_phi)])
        elif kind == 1:  # m at or near the origin
            mag, th = rng.choice([0.0, 1e-12, 1e-3]), rng.uniform(-np.pi, np.pi)
        else:
            mag, th = rng.uniform(0, 1.4 * w_ax[-1]), rng.uniform(-np.pi, np.pi)
        a, b = mag * np.cos(th), mag * np.sin(th)
        ah, bh = 0.5 * a, 0.5 * b
        cx, cy = 0.5 * w_ax[:, None] * cphi[None, :], 0.5 * w_ax[:, None] * sphi[None, :]
        d2 = (cx - ah) ** 2 + (cy - bh) ** 2
        for br in range(-(-n_w // pm.BLK_R)):
            for bc in range(-(-n_phi // pm.BLK_C)):
                r0, r1 = br * pm.BLK_R, min(br * pm.BLK_R + pm.BLK_R, n_w) - 1
                c0, c1 = bc * pm.BLK_C, min(bc * pm.BLK_C + pm.BLK_C, n_phi) - 1
                lb = pm.cell_wind_lb(ah, bh, 0.5 * w_ax[r0], 0.5 * w_ax[r1], cphi[c0], sphi[c0], cphi[c1], sphi[c1], (c1 - c0) * dphi)
                true = d2[r0:r1 + 1, c0:c1 + 1].min()
                assert lb * (1.0 - 1e-8) <= true + 1e-8 * (1.0 + ah * ah + bh * bh), (case, br, bc, lb, true)
        mh = 0.5 * mag
        assert pm.radial_lb(mh, 0.5 * w_ax[0], 0.5 * w_ax[-1]) <= d2.min() * (1 + 1e-12) + 1e-12


@pytest.mark.parametrize("kind", ["cmod5n", "noisy", "rolloff", "wrap360"])
def test_block_pyramid_finds_the_oracle_argmin(kind):
    """block_pruned_argmin == the oracle on ordinary pixels and on the ones the a-priori side cannot bound: sigma0 10..30 dB
    above the GMF (ships, land), sigma0 far below it, an a-priori wind of the wrong size or direction; with the ray's bound,
    with no bound at all, and restricted to the disc's window.  Any LUT: noisy columns, a top that rolls off, a 0..360 axis."""
    rng = np.random.default_rng({"cmod5n": 21, "noisy": 22, "rolloff": 23, "wrap360": 24}[kind])
    phi_ax = np.linspace(0, 360, 145) if kind == "wrap360" else np.linspace(0, 180, 91)
    inc_ax, w_ax = np.linspace(18, 44, 7), np.linspace(0.5, 59.5, 237)
    co = 10 * np.log10(gmf.gmf_cmod5n(inc_ax[:, None, None], w_ax[None, :, None], phi_ax[None, None, :]) + 1e-15)
    if kind == "noisy":
        co = co + 0.1 * rng.standard_normal(co.shape)
    elif kind == "rolloff":
        co = co - 0.02 * np.maximum(w_ax[None, :, None] - 24.0, 0.0) ** 2
    lco = olut.Lut(co, inc_ax, w_ax, phi_ax, "dB", "x", "co", "VV")
    p = oinv.Prepared(lco, None)
    n = 700
    inc, wt, pt = rng.uniform(18, 46, n), rng.uniform(0.5, 35, n), rng.uniform(-180, 180, n)
    lin = gmf.gmf_cmod5n(inc, wt, pt) * rng.gamma(100, 1 / 100, n)
    lin[:150] *= 10.0 ** rng.uniform(1.0, 3.0, 150)   # outliers: +10 .. +30 dB
    lin[150:200] *= 10.0 ** -rng.uniform(1.0, 3.0, 50)  # -10 .. -30 dB
    s = oinv.to_db(lin)
    anc = wt * np.exp(1j * np.deg2rad(pt)) + rng.normal(0, 1.5, n) + 1j * rng.normal(0, 1.5, n)
    anc[100:300] = rng.uniform(0, 60, 200) * np.exp(1j * rng.uniform(-np.pi, np.pi, 200))  # a-priori wind far off
    anc[300:320] *= 1e-6
    nan = np.full(n, np.nan)
    idx = cport.invert_numpy(p, inc, s, nan, nan, anc, return_idx=True, reference_layout=False)[2]
    cphi, sphi = np.cos(np.radians(phi_ax)), np.sin(np.radians(phi_ax))
    tabs = [pm.block_tables(co[i]) for i in range(len(inc_ax))]
    tabs4 = [pm.subblock_tables(co[i]) for i in range(len(inc_ax))]
    n_w, n_phi = len(w_ax), len(phi_ax)
    swept_bound, swept_free, swept_sub = [], [], []
    for i in range(n):
        ii = int(np.argmin(np.abs(inc_ax - inc[i])))
        a, b = anc[i].real, anc[i].imag
        be = abs(b) if p.phi_180 else b
        # the ray's bound as the kernels form it: any real candidate's score, here the a-priori direction's column minimum
        theta = np.degrees(np.arctan2(be, a))
        if theta < phi_ax[0]:
            theta += 360.0
        ipr = int(np.clip(np.rint((theta - phi_ax[0]) / (phi_ax[1] - phi_ax[0])), 0, n_phi - 1))
        jcol = pm.exact_J(w_ax, cphi[ipr], sphi[ipr], co[ii][:, ipr], s[i], a, be, 0.1)
        j_ub = float(jcol.min()) * (1 + 1e-9) + 1e-9
        r = pm.block_pruned_argmin(co[ii], w_ax, phi_ax, cphi, sphi, p.phi_180, s[i], a, b, 0.1, j_ub=j_ub, tables=tabs[ii])
        assert (r[0], r[1]) == (idx[i, 0], idx[i, 1]), (kind, "bound", i, r, idx[i])
        swept_bound.append(r[2])
        # round 5: the kept blocks bounded once more per quarter (sub-block tables) -- the same argmin from fewer candidates
        r4 = pm.block_pruned_argmin(co[ii], w_ax, phi_ax, cphi, sphi, p.phi_180, s[i], a, b, 0.1, j_ub=j_ub, tables=tabs[ii], sub_tables=tabs4[ii])
        assert (r4[0], r4[1]) == (idx[i, 0], idx[i, 1]), (kind, "quarters", i, r4, idx[i])
        swept_sub.append(r4[2])
        if i % 2 == 0:  # ... and with level 1 by cells of 8 x 2 blocks (k_invert_blocks) instead of bands over all directions
            rc = pm.block_pruned_argmin(co[ii], w_ax, phi_ax, cphi, sphi, p.phi_180, s[i], a, b, 0.1, j_ub=j_ub, tables=tabs[ii], sub_tables=tabs4[ii], cells=True)
            assert (rc[0], rc[1]) == (idx[i, 0], idx[i, 1]), (kind, "cells", i, rc, idx[i])
        if i % 5 == 0:  # no bound at all: the pyramid finds its own
            r = pm.block_pruned_argmin(co[ii], w_ax, phi_ax, cphi, sphi, p.phi_180, s[i], a, b, 0.1, tables=tabs[ii])
            assert (r[0], r[1]) == (idx[i, 0], idx[i, 1]), (kind, "free", i, r, idx[i])
            swept_free.append(r[2])
        if i % 3 == 0:  # inside the disc's window only
            mag = float(np.hypot(a, be))
            win = pm.search_window(mag, theta, j_ub, w_ax[0], (n_w - 1) / (w_ax[-1] - w_ax[0]), n_w, phi_ax[0], phi_ax[-1],
                                   (n_phi - 1) / (phi_ax[-1] - phi_ax[0]), n_phi)
            r = pm.block_pruned_argmin(co[ii], w_ax, phi_ax, cphi, sphi, p.phi_180, s[i], a, b, 0.1, j_ub=j_ub, window=win, tables=tabs[ii])
            assert (r[0], r[1]) == (idx[i, 0], idx[i, 1]), (kind, "window", i, r, idx[i])
    nblocks = -(-n_w // pm.BLK_R) * -(-n_phi // pm.BLK_C)
    assert np.mean(swept_bound) < 0.08 * nblocks, (np.mean(swept_bound), nblocks)
    # outliers alone (their windows are the whole grid): still a small part of the table
    assert np.mean(swept_bound[:150]) < 0.15 * nblocks, np.mean(swept_bound[:150])
    # the quarters: never more, and on the smooth GMF well under half of the candidates (outliers and far-off a-priori winds alike)
    assert all(q <= w + 1e-9 for q, w in zip(swept_sub, swept_bound))
    if kind in ("cmod5n", "rolloff", "wrap360"):
        assert np.mean(swept_sub) < 0.6 * np.mean(swept_bound), (np.mean(swept_sub), np.mean(swept_bound))


def _lut_for(kind, rng):
    inc_ax, w_ax, phi_ax = np.linspace(20, 44, 5), np.linspace(0.5, 39.5, 196), np.linspace(0, 180, 91)
    co = 10 * np.log10(gmf.gmf_cmod5n(inc_ax[:, None, None], w_ax[None, :, None], phi_ax[None, None, :]) + 1e-15)
    if kind == "plateaus":
        co = np.round(co / 0.05) * 0.05
    elif kind == "rolloff":
        co = co - 0.004 * np.maximum(w_ax[None, :, None] - 22.0, 0.0) ** 2
    return inc_ax, w_ax, phi_ax, co


@pytest.mark.parametrize("kind", ["smooth", "plateaus", "rolloff"])
def test_contour_bound_and_joint_shrink_never_exclude_the_argmin(kind):
    """Round 5 (k_invert_band2): the contour bound + per-direction joint shrink of window and band find the oracle's argmin on
    pixels whose a-priori wind is far from the sigma0 contour (x 0.3 ... x 2.5 of the truth) -- with a small fraction of the
    candidates the separate bounds leave."""
    rng = np.random.default_rng({"smooth": 11, "plateaus": 12, "rolloff": 13}[kind])
    inc_ax, w_ax, phi_ax, co = _lut_for(kind, rng)
    lco = olut.Lut(co, inc_ax, w_ax, phi_ax, "dB", "x", "co", "VV")
    p = oinv.Prepared(lco, None)
    n = 240
    inc, wt, pt = rng.uniform(21, 43, n), rng.uniform(3, 25, n), rng.uniform(-180, 180, n)
    s = oinv.to_db(gmf.gmf_cmod5n(inc, wt, pt) * rng.gamma(100, 1 / 100, n))
    scale = rng.choice([0.3, 0.6, 1.0, 1.6, 2.5], n)
    anc = scale * wt * np.exp(1j * np.deg2rad(pt)) + rng.normal(0, 1.5, n) + 1j * rng.normal(0, 1.5, n)
    nan = np.full(n, np.nan)
    idx = cport.invert_numpy(p, inc, s, nan, nan, anc, return_idx=True, reference_layout=False)[2]
    cphi, sphi = np.cos(np.radians(phi_ax)), np.sin(np.radians(phi_ax))
    swept, plain, done = [], [], 0
    for i in range(n):
        ii = np.argmin(np.abs(inc_ax - inc[i]))
        r = pm.refined_band_argmin(co[ii], w_ax, phi_ax, cphi, sphi, p.phi_180, s[i], anc[i].real, anc[i].imag, 0.1, bins=512)
        if r is None:
            continue  # (handed on: near-tie, or a tail longer than the sweep takes)
        done += 1
        assert (r[0], r[1]) == (idx[i, 0], idx[i, 1]), (kind, i, r, idx[i])
        swept.append(r[2])
        r0 = pm.refined_band_argmin(co[ii], w_ax, phi_ax, cphi, sphi, p.phi_180, s[i], anc[i].real, anc[i].imag, 0.1, bins=512, rounds=1, use_contour=False)
        if r0 is not None:
            assert (r0[0], r0[1]) == (idx[i, 0], idx[i, 1])
            plain.append(r0[2])
    assert done > 0.8 * n
    assert np.mean(swept) < 0.5 * np.mean(plain), (np.mean(swept), np.mean(plain))


@pytest.mark.parametrize("seed", range(3))
def test_joint_rows_keep_every_candidate_under_the_bound(seed):
    """`joint_rows` (steps B, A, B, A from the inverse-row table alone) never drops a row whose score is within the bound, for
    random directions, bounds and sigma0 on a monotone column; and `table_bins_margin` brackets the thresholds."""
    rng = np.random.default_rng(100 + seed)
    n_w = 160
    w_ax = np.linspace(0.4, 32.2, n_w)
    wh = 0.5 * w_ax
    w0, inv_wstep = w_ax[0], (n_w - 1) / (w_ax[-1] - w_ax[0])
    n_dead = 0
    for case in range(400):
        col = np.cumsum(rng.uniform(0.0, 0.3, n_w)) - 30.0
        if case % 5 == 0:
            col = np.round(col / 0.1) * 0.1  # plateaus
        bins = int(rng.choice([64, 512, 2048]))
        width = (col.max() - col.min()) / (bins - 2)
        grid = (col.min() - width, width, 1.0 / width)
        inv_col = pm.inverse_rows(col, n_w, grid[0], grid[1], bins)
        s = rng.uniform(col.min() - 1.0, col.max() + 1.0)
        dsig = float(rng.choice([0.1, 0.3, 1.0]))
        ah, bh = rng.uniform(-12, 12, 2)
        m2 = ah * ah + bh * bh
        ang = rng.uniform(0, np.pi)
        uh = ah * np.cos(ang) + bh * np.sin(ang)
        J = wh * (wh - 2 * uh) + m2 + ((col - s) / dsig) ** 2
        j_ub = float(rng.choice([J.min() * (1 + 1e-9) + 1e-9, np.quantile(J, rng.uniform(0, 0.3)), rng.uniform(0.1, 80.0)]))
        w_lo, w_hi = sorted(rng.integers(0, n_w, 2))
        for rounds in (1, 2, 3):
            lo, hi, _ = pm.joint_rows(inv_col, grid, s, dsig, j_ub, uh, m2, wh[0], wh[1] - wh[0], w0, inv_wstep, int(w_lo), int(w_hi), rounds)
            under = np.nonzero(J[w_lo:w_hi + 1] <= j_ub)[0] + w_lo
            if under.size:
                assert lo <= under.min() and under.max() <= hi, (case, rounds, lo, hi, under.min(), under.max())
        # the live-arc test of a direction (window_arc in stage 1 of k_invert_band, live_arc in k_invert_band2): with the pixel's own
        # band bins, a direction that holds a candidate under the bound is never declared dead
        d = np.sqrt(max(j_ub, 0.0)) * dsig * (1 + 1e-6) + 1e-9
        tb_lo, tb_hi = pm.table_bins(grid[0], width, grid[2], bins, s - d, s + d)
        live = pm.direction_is_live(inv_col, tb_lo, tb_hi if tb_hi >= 0 else bins, bins, j_ub, uh, m2, wh[0], wh[1] - wh[0], int(w_lo), int(w_hi))
        if (J[w_lo:w_hi + 1] <= j_ub).any():
            assert live, (case, "live arc", tb_lo, tb_hi)
        n_dead += 0 if live else 1
        thr_lo, thr_hi = s - rng.uniform(0, 2), s + rng.uniform(0, 2)
        b_lo, b_hi = pm.table_bins_margin(grid[0], grid[2], bins, thr_lo, thr_hi)
        assert b_lo == 0 or b_lo * width + grid[0] <= thr_lo
        assert b_hi == bins or b_hi * width + grid[0] > thr_hi
    assert n_dead > 40, n_dead  # (the test has teeth: a good part of the random directions IS declared dead)
