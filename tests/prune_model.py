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
    # (round 3: the device forms mag, sqrt(j_ub), asin and theta with float32 library calls -- stage 1 is VALU-bound -- and
    # covers their error by wider margins; a window may only ever grow.  The same margins here.)
    MRG = 2e-3
    j_ub = j_ub * (1.0 + 1e-9) + 1e-9
    R = 2.0 * np.sqrt(j_ub) * (1.0 + 1e-6) + 1e-6
    if not (mag < 1e6 and R < 1e6):
        return 0, n_w - 1, 0, n_phi - 1
    xl, xh = (mag - R - w0) * inv_wstep, (mag + R - w0) * inv_wstep
    w_lo = int(max(np.ceil(np.clip(xl - MRG - 1e-9 * abs(xl), -4.0, n_w + 4.0)), 0))
    w_hi = int(min(np.floor(np.clip(xh + MRG + 1e-9 * abs(xh), -4.0, n_w + 4.0)), n_w - 1))
    if not (R < mag * 0.999):
        return w_lo, w_hi, 0, n_phi - 1  # the disc contains the origin (or nearly: no arcsine above 0.999): every direction
    half = np.degrees(np.arcsin(R / mag)) + 2e-4
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
    """Returns (i_wspd, i_phi, n_evaluated) for one pixel; slice_wp is the (n_w, n_phi) dB slice.  (One ray: the general
    kernel takes the smallest score of three since the end of round 3, XSW_STRIP_RAYS -- any score seen is a valid bound, the
    three-ray form is the one band_pruned_argmin spells out.)"""
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


# ---------------------------------------------------------------------------------------------------------------------
# Band pruning (round 2): the sigma0 term alone is also >= 0, so a candidate with ((L - s)/dsig)^2 > J_ub cannot be the
# argmin either: |L - s| <= d = |dsig| sqrt(J_ub) is necessary.  Where every LUT column is non-decreasing in the wind
# speed over the rows of the window, that set is ONE row interval per direction, [lower_bound(s - d), upper_bound(s + d)),
# found by bisection inside the window: a handful of candidates per direction instead of the whole window column.
def mono_rows(slice_wp):
    """Largest R such that every column of the (n_w, n_phi) slice is non-decreasing over rows [0, R)."""
    dec = np.diff(slice_wp, axis=0) < 0  # also False for NaN (a NaN LUT is not prunable at all)
    if not dec.any():
        return slice_wp.shape[0]
    first = np.where(dec.any(axis=0), dec.argmax(axis=0), slice_wp.shape[0])
    return int(first.min()) + 1


def tail_min(slice_wp, mono, ip_lo=None, ip_hi=None):
    """Smallest LUT value of the rows >= mono in the directions ip_lo .. ip_hi (default: all of them); +inf when every row is
    monotone: what a query of k_tail_min's sparse table returns."""
    cols = slice(None) if ip_lo is None else slice(ip_lo, ip_hi + 1)
    return float(slice_wp[mono:, cols].min()) if mono < slice_wp.shape[0] else np.inf


def band_radius(j_ub, dsig):
    """d such that every candidate with ((L - s)/dsig)^2 <= j_ub has |L - s| <= d (inflated: rounding never excludes one)."""
    return abs(dsig) * np.sqrt(j_ub) * (1.0 + 1e-6) + 1e-9


def band_pruned_argmin(slice_wp, wspd, phi, cphi, sphi, phi_180, s, a, b, dsig, max_len=16, tail_cut=True, tail_sweep=0):
    """Like pruned_argmin, with the band rule on top of the disc window.  Returns (i_wspd, i_phi, n_evaluated, used_band);
    falls back to pruned_argmin when the window leaves the monotone rows, a band is longer than max_len, or the screening
    cannot decide (the device then re-does the pixel with the window sweep)."""
    n_w, n_phi = slice_wp.shape
    if phi_180:
        b = abs(b)
    if not (np.isfinite(s) and np.isfinite(a) and np.isfinite(b)):
        return None
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
    npairs = (n_w + 1) >> 1
    rbest, nray = np.inf, 0
    # three rays (k_invert_band: co_window_lanes<3, 2>): the direction nearest to m and the ones two grid steps to either side
    seed = 0
    for q, ipq in enumerate((ipr, max(ipr - 2, 0), min(ipr + 2, n_phi - 1))):
        ur = 2.0 * (ah * cphi[ipq] + bh * sphi[ipq])
        # the side rays bisect 3 steps in a bracket of +-4 row pairs around the first ray's result (any score seen is a bound)
        lo, hi = (0, npairs) if q == 0 else (max(seed - 4, 0), min(seed + 4, npairs))
        col = slice_wp[:, ipq]
        for _ in range(int(npairs).bit_length() if q == 0 else 3):
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
        if q == 0:
            seed = lo
    j_ub = (rbest + m2) * (1.0 + 1e-9) + 1e-9
    w_lo, w_hi, ip_lo, ip_hi = search_window(mag, theta, rbest + m2, w0, inv_wstep, n_w, phi0, phi[-1], inv_dphi, n_phi)

    def fallback():
        r = pruned_argmin(slice_wp, wspd, phi, cphi, sphi, False, s, a, b, dsig)
        return r[0], r[1], r[2], False

    d = band_radius(j_ub, dsig)
    mono = mono_rows(slice_wp)
    if w_hi >= mono and tail_cut and s + d < (tail_min(slice_wp, mono) if tail_cut == "whole" else tail_min(slice_wp, mono, ip_lo, ip_hi)):
        # Tail cut (round 3; band_wave, L.tail_min = k_tail_min at LUT install): the window reaches past the monotone rows, but
        # every LUT value of the rows >= mono in the window's directions lies above s + d, so none of those rows is in the band:
        # their sigma0 term alone exceeds J_ub.  (tail_cut="whole": the minimum over all the directions, the first form.)  The window is cut at the last monotone row and the band rule applies to what is left.
        w_hi = mono - 1
    tail_rows = []
    if w_hi >= mono and tail_sweep and mono >= 1 and w_hi - mono + 1 <= tail_sweep:
        # Tail sweep (round 3; k_invert_band2): the band does reach the rows past the monotone ones.  The band rule still holds
        # on the monotone part; the rows mono .. w_hi are all kept as candidates (every one belongs to the window).
        tail_rows = list(range(max(mono, w_lo), w_hi + 1))
        w_hi = mono - 1
    if w_hi >= mono or (w_hi < w_lo and not tail_rows) or ip_hi - ip_lo + 1 > 64:
        return fallback()
    cand = []
    for ip in range(ip_lo, ip_hi + 1):
        c = slice_wp[:, ip]
        lo_r, hi_r = w_lo, w_hi + 1
        while lo_r < hi_r:  # lower_bound(s - d) inside the window
            mid = (lo_r + hi_r) >> 1
            if c[mid] < s - d:
                lo_r = mid + 1
            else:
                hi_r = mid
        r = lo_r
        n = 0
        while r <= w_hi and c[r] <= s + d:
            n += 1
            if n > max_len:
                return fallback()
            u = 2.0 * (ah * cphi[ip] + bh * sphi[ip])
            cand.append((wh[r] * (wh[r] - u) + (c[r] * inv + sn) ** 2, r, ip))
            r += 1
        for r in tail_rows:
            u = 2.0 * (ah * cphi[ip] + bh * sphi[ip])
            cand.append((wh[r] * (wh[r] - u) + (c[r] * inv + sn) ** 2, r, ip))
    assert cand, "the ray's best candidate always lies in the band"
    js = np.array([c[0] for c in cand])
    gmin = js.min()
    keep = [c for c in cand if c[0] <= gmin + screening_eps(gmin, m2)]
    if len(keep) != 1:
        return fallback()  # the device settles near-ties in the window sweep
    return keep[0][1], keep[0][2], len(cand) + nray, True


# ---------------------------------------------------------------------------------------------------------------------
# Inverse-row table (round 2, second half): the band's rows without a search.  A monotone column is inverted once:
# inv[b] = first row r < mono with col[r] >= t_b,  t_b = fma(b, width, t0)  (else mono),  inv[0] = 0  (xsw_lutbuild.hpp:
# k_inv_range / k_inv_rows).  For a pixel, the largest t_b <= s - d gives a row at or below the band's first row, the
# smallest t_b > s + d one past a row at or above its last (co_band_pass / stage 1 of k_invert_band, xsw_band.hpp).
def inverse_rows(col, mono, t0, width, bins):
    """The table of one column: numpy restatement of k_inv_rows."""
    out = np.zeros(bins, dtype=np.int64)
    r = 0
    for b in range(1, bins):
        thr = b * width + t0  # the kernel uses fma; the rounding difference is what the in-kernel checks absorb either way
        while r < mono and col[r] < thr:
            r += 1
        out[b] = r
    return out


def table_bins(t0, width, inv_width, bins, thr_lo, thr_hi):
    """(lower bin, upper bin or -1) as stage 1 of k_invert_band picks them."""
    b = int(min(max((thr_lo - t0) * inv_width, 0.0), float(bins - 1)))
    if b > 0 and b * width + t0 > thr_lo:
        b -= 1
    if b > 0 and b * width + t0 > thr_lo:
        b = 0
    bh = int(min(max((thr_hi - t0) * inv_width, -1.0), float(bins))) + 1
    if bh < bins and not (bh * width + t0 > thr_hi):
        bh += 1
    if bh < bins and not (bh * width + t0 > thr_hi):
        bh = bins
    return b, (bh if bh < bins else -1)


def band_rows_from_table(inv, b_lo, b_hi, w_lo, w_hi):
    """[first, last] rows a lane looks at (last < first: none), as co_band_pass clips them."""
    first = max(int(inv[b_lo]), w_lo)
    last = min(int(inv[b_hi]) - 1, w_hi) if b_hi >= 0 else w_hi
    return first, last


# ---------------------------------------------------------------------------------------------------------------------
# Chord clip (round 3, k_invert_band2): the window is only the bounding box of the disc |c - m| <= 2 sqrt(J_ub).  Along one
# direction e (unit vector), with U = m . e and wh = w / 2:  |w e - m|^2 / 4 = wh^2 - U wh + |m|^2/4 <= J_ub
#   <=>  |wh - U/2| <= sqrt(U^2/4 - |m|^2/4 + J_ub):  the rows of that direction that can hold the argmin are one interval.
# The device recovers J_ub from the band's half width (thr_hi - thr_lo) / 2 = band_radius(j_ub, dsig) and takes the square
# root in float32; the same operations here.
CHORD_MRG = 2e-3


def chord_rows(ah, bh, cphi_e, sphi_e, thr_lo, thr_hi, inv_dsig, w0, inv_wstep):
    """(c_lo, c_hi) = first and last speed row of direction e inside the (inflated) disc, or None when the ray misses it.
    ah, bh = m / 2; thr_lo, thr_hi = s -+ d as the slot holds them."""
    m2 = ah * ah + bh * bh
    rs = 0.5 * (thr_hi - thr_lo) * abs(inv_dsig)  # >= sqrt(J_ub) (1 + 1e-6)
    jrel = (rs * rs - m2) + 1e-9 * (rs * rs + m2)
    uh = ah * cphi_e + bh * sphi_e
    disc = uh * uh + jrel
    if disc < 0.0:
        return None
    h = float(np.sqrt(np.float32(max(disc, 0.0)))) * (1.0 + 1e-6) + 1e-6
    inv_whs = 2.0 * inv_wstep
    xc, xh = (uh - 0.5 * w0) * inv_whs, h * inv_whs + CHORD_MRG
    return int(np.ceil(max(xc - xh, -4.0))), int(np.floor(min(xc + xh, 40000.0)))


# ---------------------------------------------------------------------------------------------------------------------
# Block pyramid (round 4, `co_block_search` of the general kernel): a bound from the TABLE side, for the pixels whose a-priori
# wind does not confine the search (sigma0 outliers: windows covering the whole grid; windows past the monotone rows with long
# runs).  The slice is cut into blocks of BLK_R speed rows x BLK_C directions; per block the smallest and largest LUT value
# (float32, rounded outward; xsw_lutbuild.hpp: k_block_minmax) and per BAND of `band_rows(n_phi)` block rows the same over all
# directions (k_band_minmax).  For a pixel,
#     LB(block) = (max(0, lo - s, s - hi) / dsig)^2 + (distance of m/2 to the block's polar cell in half-speed units)^2
# is a lower bound of J = Jsig + Jwind over the block's candidates: BOTH terms bound together, where the window (disc) and the
# band rule bound each term separately.  Any LUT (no monotonicity), any s.  A block is skipped when LB (deflated) exceeds the
# best score known (the ray's J_ub, then the running minimum of the sweep); a candidate of a skipped block scores strictly
# above an examined one in the reference's arithmetic, so it cannot be the argmin and cannot tie with it.
BLK_R, BLK_C = 4, 16


def f32_down(x):
    y = np.float32(x)
    return y if float(y) <= x else np.nextafter(y, np.float32(-np.inf))


def f32_up(x):
    y = np.float32(x)
    return y if float(y) >= x else np.nextafter(y, np.float32(np.inf))


def block_tables(slice_wp):
    """(lo, hi) float32 [nbr, nbc]: outward-rounded min / max of every BLK_R x BLK_C block of the (n_w, n_phi) slice."""
    n_w, n_phi = slice_wp.shape
    nbr, nbc = -(-n_w // BLK_R), -(-n_phi // BLK_C)
    lo, hi = np.empty((nbr, nbc), np.float32), np.empty((nbr, nbc), np.float32)
    for br in range(nbr):
        for bc in range(nbc):
            blk = slice_wp[br * BLK_R:(br + 1) * BLK_R, bc * BLK_C:(bc + 1) * BLK_C]
            lo[br, bc], hi[br, bc] = f32_down(float(blk.min())), f32_up(float(blk.max()))
    return lo, hi


def band_rows(n_phi):
    """Block rows per band = the block rows one wave trip of 64 lanes covers (lane = block): 64 // nbc, at least 1."""
    return max(1, 64 // (-(-n_phi // BLK_C)))


def band_tables(lo, hi, n_phi):
    g = band_rows(n_phi)
    nb = -(-lo.shape[0] // g)
    return (np.array([lo[t * g:(t + 1) * g].min() for t in range(nb)], np.float32),
            np.array([hi[t * g:(t + 1) * g].max() for t in range(nb)], np.float32))


def sig_lb(lo, hi, s, inv_dsig):
    d = max(0.0, float(lo) - s, s - float(hi))
    return (d * abs(inv_dsig)) ** 2


def radial_lb(mh, wha, whb):
    """(| |c|/2 - |m|/2 |)^2 <= |c - m|^2 / 4 for every c with wha <= |c|/2 <= whb."""
    r = max(0.0, wha - mh, mh - whb)
    return r * r


def cell_wind_lb(ah, bh, wha, whb, ca, sa, cb, sb, span_deg):
    """Lower bound of |c/2 - m/2|^2 over the polar cell  wha <= |c|/2 <= whb,  direction between the unit vectors (ca, sa) and
    (cb, sb) (counter-clockwise from a to b, span_deg < 170): the radial bound, and -- when m lies outside the cell's angular
    sector -- the squared distance to the nearer edge segment.  m = 2 (ah, bh)."""
    m2 = ah * ah + bh * bh
    mh = np.sqrt(m2)
    lb = radial_lb(mh, wha, whb)
    if not (span_deg < 170.0):
        return lb
    tol = 1e-9 * (mh + 1e-300)
    inside = (ca * bh - sa * ah >= -tol) and (ah * sb - bh * cb >= -tol)  # cross(ea, m) >= 0 and cross(m, eb) >= 0
    if inside:
        return lb
    pm = max(ah * ca + bh * sa, ah * cb + bh * sb)  # |m/2| cos(angle to the nearer edge)
    t = min(max(pm, wha), whb)
    return max(lb, m2 + t * (t - 2.0 * pm))


def block_keep(lb, j_ub, m2):
    """A block (band) is examined unless its lower bound, deflated for the rounding of its own arithmetic, exceeds the bound."""
    return not (lb * (1.0 - 1e-8) > j_ub + 1e-8 * (1.0 + m2))


BLK_C4 = 4  # directions of a SUB-BLOCK (round 5, k_invert_blocks): a kept block is bounded once more per quarter before it is swept


def subblock_tables(slice_wp):
    """(lo, hi) float32 [nbr, nbc4]: block_tables per sub-block of BLK_R speed rows x BLK_C4 directions."""
    n_w, n_phi = slice_wp.shape
    nbr, nbc4 = -(-n_w // BLK_R), -(-n_phi // BLK_C4)
    lo, hi = np.empty((nbr, nbc4), np.float32), np.empty((nbr, nbc4), np.float32)
    for br in range(nbr):
        for bc in range(nbc4):
            blk = slice_wp[br * BLK_R:(br + 1) * BLK_R, bc * BLK_C4:(bc + 1) * BLK_C4]
            lo[br, bc], hi[br, bc] = f32_down(float(blk.min())), f32_up(float(blk.max()))
    return lo, hi


CELL_R, CELL_C = 8, 2  # blocks per level-1 CELL of k_invert_blocks (round 5): 32 speed rows x 32 directions


def cell_tables(lo, hi):
    """(lo, hi) float32 [ncr, ncc] of the cells of CELL_R x CELL_C blocks, from block_tables' arrays."""
    nbr, nbc = lo.shape
    ncr, ncc = -(-nbr // CELL_R), -(-nbc // CELL_C)
    clo, chi = np.empty((ncr, ncc), np.float32), np.empty((ncr, ncc), np.float32)
    for i in range(ncr):
        for j in range(ncc):
            clo[i, j] = lo[i * CELL_R:(i + 1) * CELL_R, j * CELL_C:(j + 1) * CELL_C].min()
            chi[i, j] = hi[i * CELL_R:(i + 1) * CELL_R, j * CELL_C:(j + 1) * CELL_C].max()
    return clo, chi


def block_pruned_argmin(slice_wp, wspd, phi, cphi, sphi, phi_180, s, a, b, dsig, j_ub=np.inf, window=None, tables=None, sub_tables=None, cells=False):
    """The reference's argmin of one pixel by the block pyramid.  j_ub: any valid upper bound of the minimum of J (inf: none);
    window = (w_lo, w_hi, ip_lo, ip_hi): only blocks that touch it are looked at (the disc's bounding box; None: the grid).
    sub_tables = subblock_tables(slice): a kept block is not swept whole -- each of its quarters (BLK_C4 directions) is bounded
    once more from its own {min, max} and its own, narrower polar cell, and only the quarters that survive are swept (round 5:
    where the GMF saturates sigma0 varies faster with the direction than with the speed, a block 16 directions wide nearly always
    straddles the contour and its sigma0 bound is zero).
    cells = True: level 1 is k_invert_blocks's (round 5) -- cells of CELL_R x CELL_C blocks, each with its own {min, max} AND its own
    polar cell (a band over all directions has a sigma0 range that holds nearly any s, and only the radial wind bound), the most
    promising cell first, then every cell the tightened bound keeps.
    Returns (i_wspd, i_phi, blocks swept -- quarters count as 1/4 --, bands (cells) kept)."""
    n_w, n_phi = slice_wp.shape
    if phi_180:
        b = abs(b)
    lo, hi = tables if tables is not None else block_tables(slice_wp)
    blo, bhi = band_tables(lo, hi, n_phi)
    g = band_rows(n_phi)
    nbr, nbc = lo.shape
    inv = 1.0 / dsig
    sn = -s * inv
    ah, bh = 0.5 * a, 0.5 * b
    m2 = ah * ah + bh * bh
    mh = np.sqrt(m2)
    wh = 0.5 * wspd
    dphi = (phi[-1] - phi[0]) / (n_phi - 1)
    w_lo, w_hi, ip_lo, ip_hi = window if window is not None else (0, n_w - 1, 0, n_phi - 1)
    br_lo, br_hi, bc_lo, bc_hi = w_lo // BLK_R, w_hi // BLK_R, ip_lo // BLK_C, ip_hi // BLK_C

    def lb_of(br, bc):
        r0, r1 = br * BLK_R, min(br * BLK_R + BLK_R, n_w) - 1
        c0, c1 = bc * BLK_C, min(bc * BLK_C + BLK_C, n_phi) - 1
        return sig_lb(lo[br, bc], hi[br, bc], s, inv) + cell_wind_lb(ah, bh, wh[r0], wh[r1], cphi[c0], sphi[c0], cphi[c1], sphi[c1],
                                                                       (c1 - c0) * dphi)

    cand = {}  # flat -> screening score
    state = {"jub": j_ub, "swept": 0, "bands": 0}

    def sweep_cols(br, c_from, c_to):
        for r in range(br * BLK_R, min(br * BLK_R + BLK_R, n_w)):
            for c in range(c_from, c_to):
                u = 2.0 * (ah * cphi[c] + bh * sphi[c])
                js = wh[r] * (wh[r] - u) + (slice_wp[r, c] * inv + sn) ** 2
                cand[r * n_phi + c] = js
                state["jub"] = min(state["jub"], (js + m2) * (1.0 + 1e-9) + 1e-9)

    def sweep(br, bc):
        if sub_tables is None:
            state["swept"] += 1
            sweep_cols(br, bc * BLK_C, min(bc * BLK_C + BLK_C, n_phi))
            return
        lo4, hi4 = sub_tables
        r0, r1 = br * BLK_R, min(br * BLK_R + BLK_R, n_w) - 1
        for bc4 in range(bc * (BLK_C // BLK_C4), (bc + 1) * (BLK_C // BLK_C4)):
            c0 = bc4 * BLK_C4
            if c0 >= n_phi:
                break
            c1 = min(c0 + BLK_C4, n_phi) - 1
            lb4 = sig_lb(lo4[br, bc4], hi4[br, bc4], s, inv) + cell_wind_lb(ah, bh, wh[r0], wh[r1], cphi[c0], sphi[c0], cphi[c1], sphi[c1],
                                                                                (c1 - c0) * dphi)
            if block_keep(lb4, state["jub"], m2):
                state["swept"] += BLK_C4 / BLK_C
                sweep_cols(br, c0, c1 + 1)

    if cells:
        clo, chi = cell_tables(lo, hi)

        def cell_lb(i, j):
            r0, r1 = i * CELL_R * BLK_R, min((i + 1) * CELL_R * BLK_R, n_w) - 1
            c0, c1 = j * CELL_C * BLK_C, min((j + 1) * CELL_C * BLK_C, n_phi) - 1
            return sig_lb(clo[i, j], chi[i, j], s, inv) + cell_wind_lb(ah, bh, wh[r0], wh[r1], cphi[c0], sphi[c0], cphi[c1], sphi[c1], (c1 - c0) * dphi)

        def do_cell(i, j):
            state["bands"] += 1
            for br in range(max(i * CELL_R, br_lo), min((i + 1) * CELL_R, nbr, br_hi + 1)):
                for bc in range(max(j * CELL_C, bc_lo), min((j + 1) * CELL_C, nbc, bc_hi + 1)):
                    if block_keep(lb_of(br, bc), state["jub"], m2):
                        sweep(br, bc)

        todo = [(i, j) for i in range(br_lo // CELL_R, br_hi // CELL_R + 1) for j in range(bc_lo // CELL_C, bc_hi // CELL_C + 1)]
        first = min(todo, key=lambda ij: cell_lb(*ij))
        do_cell(*first)
        for ij in todo:
            if ij != first and block_keep(cell_lb(*ij), state["jub"], m2):
                do_cell(*ij)
        todo_bands = ()
    else:
        todo_bands = range(br_lo // g, br_hi // g + 1)
    for t in todo_bands:
        rows0, rows1 = t * g * BLK_R, min((t + 1) * g * BLK_R, n_w) - 1
        lb1 = sig_lb(blo[t], bhi[t], s, inv) + radial_lb(mh, wh[rows0], wh[rows1])
        if not block_keep(lb1, state["jub"], m2):
            continue
        state["bands"] += 1
        for br in range(max(t * g, br_lo), min((t + 1) * g, nbr, br_hi + 1)):
            for bc in range(bc_lo, bc_hi + 1):
                if block_keep(lb_of(br, bc), state["jub"], m2):
                    sweep(br, bc)
    assert cand, "nothing examined: j_ub was not an upper bound of the minimum"
    gmin = min(cand.values())
    T = gmin + screening_eps(gmin, m2)
    best = None
    for flat, js in cand.items():
        if js <= T:
            iw, ip = divmod(flat, n_phi)
            key = (exact_J(wspd[iw], cphi[ip], sphi[ip], slice_wp[iw, ip], s, a, b, dsig), flat)
            if best is None or key < best:
                best = key
    return best[1] // n_phi, best[1] % n_phi, state["swept"], state["bands"]


# ---------------------------------------------------------------------------------------------------------------------
# Round 5, k_invert_band2 (and the entry of k_invert_blocks): the pixels whose band holds long runs have an a-priori wind far
# from the sigma0 contour.  Two refinements, both exact (they only ever drop candidates that score above a real one):
#
# CONTOUR BOUND (`contour_bound`).  The three rays of stage 1 look where the a-priori wind points; the minimum of J lies where
# the CONTOUR LUT = s comes closest to m, which may be tens of degrees away.  The inverse-row table gives the contour's row in
# any direction with one read: a coarse scan over the window's directions (<= CONTOUR_PROBES of them, then two halvings of the
# stride around the best) scores the two rows around the crossing; the smallest score is a valid upper bound (every probe is a
# real candidate), and in practice IS the minimum to a few percent.
#
# JOINT SHRINK (`joint_rows`).  Window and band bound each cost term by J_ub on its own; per direction both hold TOGETHER:
#   J(r) <= J_ub  =>  Jsig(r) <= J_ub - min Jwind over the rows still in question      (B: a narrower band, from the table)
#                 =>  Jwind(r) <= J_ub - min Jsig over the rows still in question      (A: a shorter chord, analytic)
# min Jwind over a row interval is analytic (a parabola in w/2); min Jsig over the rows between two table thresholds follows from
# the thresholds themselves (rows >= inv[b] have LUT >= t_b; rows < inv[b'] have LUT < t_b').  B, A, B, A: two table reads per
# B step, no LUT read at all; the rows that survive are swept and settled as before.
# (The device evaluates the BOUND arithmetic of these steps -- budgets, square roots, bins, chord ends -- in float32 with every result
# widened by 4e-6 of the magnitudes that went into it, xsw_band2.hpp: Bound32; the float64 forms below describe the same sets, a
# hair narrower: device rows are a superset of these, which are a superset of the rows that can hold the argmin.)
CONTOUR_PROBES = 16


def contour_bound(slice_wp, inv_tab, grid, mono, wh, cphi, sphi, s, ah, bh, inv_dsig, ip_lo, ip_hi, w_lo, w_hi):
    """Smallest screening score J_s = wh (wh - U) + ((L - s)/dsig)^2 among the rows around the crossing LUT = s of the scanned
    directions (rows clamped into the window's monotone part [w_lo, min(w_hi, mono - 1)]); (score, probes).  inv_tab[ip][b],
    grid = (t0, width, inv_width)."""
    t0, width, inv_width = grid
    bins = inv_tab.shape[1]
    b = int(min(max((s - t0) * inv_width, 0.0), float(bins - 1)))
    r_top = min(w_hi, mono - 1)
    if r_top < w_lo:
        return np.inf, 0
    sn = -s * inv_dsig

    def probe(ip):
        r0 = int(inv_tab[ip][b])
        best = np.inf
        for r in (min(max(r0 - 1, w_lo), r_top), min(max(r0, w_lo), r_top)):
            u = 2.0 * (ah * cphi[ip] + bh * sphi[ip])
            best = min(best, wh[r] * (wh[r] - u) + (slice_wp[r, ip] * inv_dsig + sn) ** 2)
        return best

    ncols = ip_hi - ip_lo + 1
    stride = max(1, -(-ncols // CONTOUR_PROBES))
    dirs = list(range(ip_lo, ip_hi + 1, stride))
    vals = [probe(ip) for ip in dirs]
    n = len(dirs)
    k = int(np.argmin(vals))
    best_ip, jc = dirs[k], vals[k]
    h = stride
    while h > 1:
        h = (h + 1) // 2
        for ip in (best_ip - h, best_ip + h):
            if ip_lo <= ip <= ip_hi:
                v = probe(ip)
                n += 1
                if v < jc:
                    jc, best_ip = v, ip
    return jc, n


def table_bins_margin(t0, inv_width, bins, thr_lo, thr_hi):
    """(lower bin, upper bin or `bins`) WITH A BIN OF MARGIN, as k_invert_band2 picks them per direction (xsw_band2.hpp:
    bins_margin): t_(b_lo) <= thr_lo - width + rounding, t_(b_hi) >= thr_hi + width - rounding -- no exact search of the bin needed."""
    b_lo = max(int(min(max((thr_lo - t0) * inv_width, 0.0), float(bins - 1))) - 1, 0)
    b_hi = min(max(int(np.floor(min(max((thr_hi - t0) * inv_width, -3.0), float(bins)))) + 2, 0), bins)
    return b_lo, b_hi


def _chord_from_budget(uh, m2, bud, w0, inv_wstep):
    """Rows with wh^2 - 2 uh wh + m2 <= bud (uh = U/2), as chord_rows computes them (float32 root, inflated); None: no row."""
    disc = uh * uh + ((bud - m2) + 1e-9 * (abs(bud) + m2))
    if disc < 0.0:
        return None
    h = float(np.sqrt(np.float32(max(disc, 0.0)))) * (1.0 + 1e-6) + 1e-6
    inv_whs = 2.0 * inv_wstep
    xc, xh = (uh - 0.5 * w0) * inv_whs, h * inv_whs + CHORD_MRG
    return int(np.ceil(max(xc - xh, -4.0))), int(np.floor(min(xc + xh, 40000.0)))


def direction_is_live(inv_col, b_lo, b_hi, bins, j_ub, uh, m2, wh0, whs, w_lo, w_hi):
    """LIVE ARC test of one direction (stage 1 of k_invert_band: window_arc; k_invert_band2: live_arc): the rows of the band
    [inv_col[b_lo], inv_col[b_hi]) inside [w_lo, w_hi] (b_lo / b_hi: the pixel's threshold bins, table_bins; b_hi = bins: no
    threshold above the band) -- the direction is dead when there is none, or when the smallest wind term the parabola
    wh^2 - 2 uh wh + m2 takes over them (its minimum clamped into the interval, deflated by the float32 slack of the device)
    exceeds the bound.  A window narrowed to [first live, last live] direction holds every candidate with J <= j_ub."""
    lo = max(w_lo, int(inv_col[b_lo]))
    hi = min(w_hi, int(inv_col[b_hi]) - 1) if b_hi < bins else w_hi
    if lo > hi:
        return False
    t = min(max(uh, wh0 + lo * whs), wh0 + hi * whs)
    jw = (m2 + t * (t - 2.0 * uh)) - 4e-6 * (m2 + t * (t + 2.0 * abs(uh)))
    return not (jw > j_ub * (1.0 + 1e-5) + 1e-5)


def joint_rows(inv_col, grid, s, dsig, j_ub, uh, m2, wh0, whs, w0, inv_wstep, w_lo, w_hi, rounds=2):
    """Rows [lo, hi] of ONE direction (monotone part of the window, [w_lo, w_hi]) that can hold a candidate with J <= j_ub
    (j_ub already inflated): steps B, A, B, A, ... as described above.  inv_col[b] = the direction's inverse-row table.
    Returns (lo, hi) (lo > hi: none), table reads."""
    t0, width, inv_width = grid
    bins = len(inv_col)
    lo, hi = w_lo, w_hi
    vlo, vhi = -np.inf, np.inf
    reads = 0
    inv_dsig = 1.0 / abs(dsig)
    for _ in range(rounds):
        if lo > hi:
            break
        # B: the smallest wind term over [lo, hi] (the parabola's own minimum clamped into the interval: a lower bound of the discrete one)
        wa, wb = wh0 + lo * whs, wh0 + hi * whs
        t = min(max(uh, wa), wb)
        jw_lb = (m2 + t * (t - 2.0 * uh)) * (1.0 - 1e-9) - 1e-9 * m2
        bud = j_ub - jw_lb
        if bud < 0.0:
            return 1, 0, reads
        d = float(np.sqrt(np.float32(max(bud, 0.0)))) * (1.0 + 1e-6) * abs(dsig) + 1e-9
        b_lo, b_hi = table_bins_margin(t0, inv_width, bins, s - d, s + d)
        reads += 2
        lo = max(lo, int(inv_col[b_lo]))
        if b_hi < bins:
            hi = min(hi, int(inv_col[b_hi]) - 1)
        vlo = (b_lo * width + t0) if b_lo > 0 else -np.inf
        vhi = (b_hi * width + t0) if b_hi < bins else np.inf
        if lo > hi:
            break
        # A: the smallest sigma0 term over rows whose LUT value lies in [vlo, vhi)
        dmin = max(0.0, vlo - s, s - vhi) * inv_dsig * (1.0 - 1e-9)
        c = _chord_from_budget(uh, m2, j_ub - dmin * dmin, w0, inv_wstep)
        if c is None:
            return 1, 0, reads
        lo, hi = max(lo, c[0]), min(hi, c[1])
    return lo, hi, reads


def refined_band_argmin(slice_wp, wspd, phi, cphi, sphi, phi_180, s, a, b, dsig, rounds=2, bins=2048, tail_sweep=256, use_contour=True):
    """The reference's argmin of one pixel the way k_invert_band2 finds it in round 5: three-ray bound, contour bound, window,
    then per direction the joint shrink on the monotone rows (+ the rows past them, clipped to the chord) and the sweep.
    Returns (i_wspd, i_phi, candidates swept, probes + table reads), or None where the kernel hands the pixel on (window past
    the monotone rows by more than tail_sweep, non-finite input, near-tie)."""
    n_w, n_phi = slice_wp.shape
    if phi_180:
        b = abs(b)
    if not (np.isfinite(s) and np.isfinite(a) and np.isfinite(b)):
        return None
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
    wh0, whs = wh[0], 0.5 / inv_wstep
    rbest = min(float(np.min(wh * (wh - 2.0 * (ah * cphi[q] + bh * sphi[q])) + (slice_wp[:, q] * inv + sn) ** 2))
                for q in (ipr, max(ipr - 2, 0), min(ipr + 2, n_phi - 1)))  # (the rays' own minima: what the bisections find on unimodal columns)
    mono = mono_rows(slice_wp)
    lo_v, hi_v = float(slice_wp[:mono].min()), float(slice_wp[:mono].max())
    width = (hi_v - lo_v) / (bins - 2) if hi_v > lo_v else 1.0
    grid = (lo_v - width, width, 1.0 / width)
    inv_tab = np.stack([inverse_rows(slice_wp[:, ip], mono, grid[0], grid[1], bins) for ip in range(n_phi)])
    w_lo, w_hi, ip_lo, ip_hi = search_window(mag, theta, rbest + m2, w0, inv_wstep, n_w, phi0, phi[-1], inv_dphi, n_phi)
    work = 0
    if use_contour:
        jc, work = contour_bound(slice_wp, inv_tab, grid, mono, wh, cphi, sphi, s, ah, bh, inv, ip_lo, ip_hi, w_lo, w_hi)
        if jc < rbest:
            rbest = jc
            w_lo, w_hi, ip_lo, ip_hi = search_window(mag, theta, rbest + m2, w0, inv_wstep, n_w, phi0, phi[-1], inv_dphi, n_phi)
    j_ub = (rbest + m2) * (1.0 + 1e-9) + 1e-9
    tail_n = 0
    if w_hi >= mono:
        if mono < 1 or w_hi - mono + 1 > tail_sweep:
            return None
        tail_n, w_hi = w_hi - mono + 1, mono - 1
    cand = []
    for ip in range(ip_lo, ip_hi + 1):
        uh = ah * cphi[ip] + bh * sphi[ip]
        u = 2.0 * uh
        rows = []
        if w_hi >= w_lo:
            lo, hi, reads = joint_rows(inv_tab[ip], grid, s, dsig, j_ub, uh, m2, wh0, whs, w0, inv_wstep, w_lo, w_hi, rounds)
            work += reads
            rows += list(range(lo, hi + 1))
        if tail_n:
            # the rows past the monotone ones: every LUT value up there is >= the direction's tail minimum (L.tail_min, level 0), so
            # their sigma0 term is at least that far from s -- the chord of the tail follows from what is left of the bound
            dmin = max(0.0, float(slice_wp[mono:, ip].min()) - s) * abs(inv) * (1.0 - 1e-9)
            c = _chord_from_budget(uh, m2, j_ub - dmin * dmin, w0, inv_wstep)
            if c is not None:
                rows += list(range(max(mono, w_lo, c[0]), min(mono + tail_n - 1, c[1]) + 1))
        for r in rows:
            cand.append((wh[r] * (wh[r] - u) + (slice_wp[r, ip] * inv + sn) ** 2, r, ip))
    assert cand, "the bound's own candidate always survives"
    gmin = min(c[0] for c in cand)
    keep = [c for c in cand if c[0] <= gmin + screening_eps(gmin, m2)]
    if len(keep) != 1:
        return None
    return keep[0][1], keep[0][2], len(cand), work
