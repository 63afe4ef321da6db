"""TEST-SIDE executable model of the device search strategy (exact branch-and-bound).

It lives under tests/ (never imported by the product) and is the specification of what the HIP kernel
`xsw_invert_pruned` does per pixel, written with numpy so that the search-window logic (the part
that decides which candidates may be skipped) can be property-tested on CPU against the oracle
(`tests/test_prune_model.py`).  The kernel in `csrc/xsw_invert.hip` follows it step by step.

Per pixel (co-pol), with m = (a, b) the ancillary wind (b := |b| for a 0..180 LUT), s the
observed sigma0 in dB, c = w*(cos phi, sin phi) a candidate and L its LUT value:

    J(c) = |c - m|^2 / 4 + ((L - s)/dsig)^2          (windspeed.py:220-225)

1. Upper bound: along the ray phi_r nearest to the direction of m, J is (nearly always) unimodal in the
   speed: bisect on the sign of its discrete slope over aligned row pairs; J_ub = the smallest score seen
   (any subset of candidates bounds the minimum from above; a column that is not unimodal only loosens
   the bound).
2. Since both terms are >= 0, a candidate with |c - m|^2/4 > J_ub cannot be the argmin: only the
   disc |c - m| <= R = 2*sqrt(J_ub) matters.  Its polar bounding box is
   w in [|m|-R, |m|+R],  phi in [theta - asin(R/|m|), theta + asin(R/|m|)]  (all phi if R >= |m|),
   taken in index space with ceil/floor and 1e-5 index units of slack.
3. Every candidate in the box is scored with a cheap float64 screening form
   J_s = wh*(wh - U_phi) + ((L*inv) + sn)^2 (+ const), wh = w/2, U_phi = a*cos + b*sin.
4. Candidates within eps of the screening minimum are re-evaluated in the reference's exact
   operation order; lowest flat index wins ties (numpy argmin).
"""
import numpy as np


def screening_eps(gmin, m2):
    return 1e-9 * (1.0 + abs(gmin) + m2)


def search_window(mag, theta_deg, j_ub, w0, inv_wstep, n_w, phi0, phi_last, inv_dphi, n_phi):
    """Index box [w_lo, w_hi] x [ip_lo, ip_hi] guaranteed to contain every candidate with
    |c - m|^2/4 <= j_ub (uniform axes).  Float64; MRG index units of slack cover the 1e-6-of-a-step
    axis tolerance, the 1e-12 trig tables and the arithmetic.  theta_deg is the direction of m
    normalised into [phi0, phi0 + 360)."""
    MRG = 1e-5
    j_ub = j_ub * (1.0 + 1e-9) + 1e-9
    R = 2.0 * np.sqrt(j_ub) * (1.0 + 1e-9) + 1e-9
    if not (mag < 1e6 and R < 1e6):
        return 0, n_w - 1, 0, n_phi - 1
    xl, xh = (mag - R - w0) * inv_wstep, (mag + R - w0) * inv_wstep
    w_lo = int(max(np.ceil(np.clip(xl - MRG - 1e-9 * abs(xl), -4.0, n_w + 4.0)), 0))
    w_hi = int(min(np.floor(np.clip(xh + MRG + 1e-9 * abs(xh), -4.0, n_w + 4.0)), n_w - 1))
    if not (R < mag * (1.0 - 1e-9)):
        return w_lo, w_hi, 0, n_phi - 1  # the disc contains the origin: every direction
    half = np.degrees(np.arcsin(R / mag)) + 1e-7
    yl, yh = (theta_deg - half - phi0) * inv_dphi, (theta_deg + half - phi0) * inv_dphi
    yl, yh = yl - MRG - 1e-9 * abs(yl), yh + MRG + 1e-9 * abs(yh)
    plo = np.ceil(np.clip(yl, -4.0, n_phi + 4.0))
    phi_hi = np.floor(np.clip(yh, -4.0, n_phi + 4.0))
    if phi_last - theta_deg <= 179.9 and theta_deg - phi0 <= 179.9:
        # no axis direction is more than 180 deg from theta: |phi - theta| is the true angular
        # distance, so directions outside the window are outside the disc -> clamp to the axis
        return w_lo, w_hi, int(max(plo, 0)), int(min(phi_hi, n_phi - 1))
    if yl >= 0 and yh <= n_phi - 1:  # the whole (unrounded) window lies on the axis: no seam inside
        return w_lo, w_hi, int(plo), int(phi_hi)
    return w_lo, w_hi, 0, n_phi - 1  # window crosses the axis seam: take every direction


def exact_J(w, cphi, sphi, lut_val, s, a, b, dsig):
    """Reference operation order (windspeed.py:220-225)."""
    jw = ((w * cphi - a) / 2) ** 2 + ((w * sphi - b) / 2) ** 2
    return jw + ((lut_val - s) / dsig) ** 2


def pruned_argmin(slice_wp, wspd, phi, cphi, sphi, phi_180, s, a, b, dsig):
    """Returns (i_wspd, i_phi, n_evaluated) for one pixel; slice_wp is the (n_w, n_phi) dB slice."""
    n_w, n_phi = slice_wp.shape
    if phi_180:
        b = abs(b)
    if not (np.isfinite(s) and np.isfinite(a) and np.isfinite(b)):
        return None  # device: exact full scan
    w0, inv_wstep = wspd[0], (n_w - 1) / (wspd[-1] - wspd[0])
    phi0, inv_dphi = phi[0], (n_phi - 1) / (phi[-1] - phi[0])
    inv = 1.0 / dsig
    sn = -s * inv
    ah, bh = 0.5 * a, 0.5 * b
    m2 = ah * ah + bh * bh
    mag = np.sqrt(a * a + b * b)
    theta = np.degrees(np.arctan2(b, a))
    if theta < phi0:
        theta += 360.0
    ipr = int(np.clip(np.rint((theta - phi0) * inv_dphi), 0, n_phi - 1))
    wh = 0.5 * wspd
    ur = 2.0 * (ah * cphi[ipr] + bh * sphi[ipr])
    npairs = (n_w + 1) >> 1
    lo, hi, rbest, nray = 0, npairs, np.inf, 0
    col = slice_wp[:, ipr]
    for _ in range(int(npairs).bit_length()):  # the kernel's fixed trip count
        mid = min((lo + hi) >> 1, npairs - 1)
        ja = wh[2 * mid] * (wh[2 * mid] - ur) + (col[2 * mid] * inv + sn) ** 2
        jb = wh[2 * mid + 1] * (wh[2 * mid + 1] - ur) + (col[2 * mid + 1] * inv + sn) ** 2 if 2 * mid + 1 < n_w else np.inf
        rbest = min(rbest, ja, jb)
        nray += 2
        if lo < hi:
            if jb < ja:
                lo = mid + 1
            else:
                hi = mid
    j_ub = rbest + m2
    w_lo, w_hi, ip_lo, ip_hi = search_window(mag, theta, j_ub, w0, inv_wstep, n_w, phi0, phi[-1], inv_dphi, n_phi)
    u = 2.0 * (ah * cphi[ip_lo:ip_hi + 1] + bh * sphi[ip_lo:ip_hi + 1])
    whb = wh[w_lo:w_hi + 1, None]
    dd = slice_wp[w_lo:w_hi + 1, ip_lo:ip_hi + 1] * inv + sn
    js = whb * (whb - u[None, :]) + dd * dd
    gmin = js.min()
    rr, cc = np.nonzero(js <= gmin + screening_eps(gmin, m2))
    best = None
    for r, c in zip(rr, cc):
        iw, ip = w_lo + r, ip_lo + c
        je = exact_J(wspd[iw], cphi[ip], sphi[ip], slice_wp[iw, ip], s, a, b, dsig)
        key = (je, iw * n_phi + ip)
        if best is None or key < best:
            best = key
    flat = best[1]
    return flat // n_phi, flat % n_phi, js.size + nray
