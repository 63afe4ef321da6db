"""Oracle restatement of the reference's analytic GMFs (test infrastructure).

Follows, operation for operation, `/root/reference/src/xsarsea/windspeed/gmfs_impl.py`:
  * CMOD5 family ........ gmfs_impl.py:8-203  (coeffs :23-55 / :59-91, body :117-163,
                          ZhangA PR :165-172, Mouche PR :174-199)
  * CMOD-IFR2 ........... gmfs_impl.py:213-303
  * VH "sum" family ..... gmf_rs2_v2 :325-385, gmf_s1_v2 :388-448, gmf_rcm_noaa :451-515
  * VH "dB-blend" family  gmf_s1_v3_ew_rec :518-551, gmf_rs2_v3 :554-589, gmf_rcm_v3 :592-628,
                          gmf_rcm_v4 :631-667, gmf_rs2_v4 :670-707

All functions are elementwise on broadcastable float64 arrays (the reference's
are scalar functions that numba vectorises, gmfs.py:202-236).  Coefficients
are physical constants of the published GMFs.
"""
import numpy as np

# --- CMOD5 / CMOD5.N coefficient sets (index 0 unused, as in the published tables) ---
_C_CMOD5 = (
    0.0, -0.688, -0.793, 0.338, -0.173, 0.0, 0.004, 0.111, 0.0162, 6.34, 2.57, -2.18, 0.4, -0.6,
    0.045, 0.007, 0.33, 0.012, 22.0, 1.95, 3.0, 8.39, -3.44, 1.36, 5.35, 1.99, 0.29, 3.80, 1.53,
)
_C_CMOD5N = (
    0.0, -0.6878, -0.7957, 0.338, -0.1728, 0.0, 0.004, 0.1103, 0.0159, 6.7329, 2.7713, -2.2885,
    0.4971, -0.725, 0.045, 0.0066, 0.3222, 0.012, 22.7, 2.0813, 3.0, 8.3659, -3.3428, 1.3236,
    6.2437, 2.3893, 0.3249, 4.159, 1.693,
)


def _cmod5_core(c, inc, wspd, phi):
    """gmfs_impl.py:117-163."""
    inc, wspd, phi = (np.asarray(v, dtype=np.float64) for v in (inc, wspd, phi))
    zpow = 1.6
    thetm = 40.0
    thethr = 25.0
    y0 = c[19]
    pn = c[20]
    a = y0 - (y0 - 1.0) / pn
    b = 1.0 / (pn * (y0 - 1.0) ** (pn - 1.0))

    cosphi = np.cos(np.deg2rad(phi))
    x = (inc - thetm) / thethr
    x2 = x ** 2.0

    a0 = c[1] + c[2] * x + c[3] * x2 + c[4] * x * x2
    a1 = c[5] + c[6] * x
    a2 = c[7] + c[8] * x
    gam = c[9] + c[10] * x + c[11] * x2
    s0 = c[12] + c[13] * x
    s = a2 * wspd
    a3_s0 = 1.0 / (1.0 + np.exp(-s0))
    with np.errstate(all="ignore"):
        low = a3_s0 * (s / s0) ** (s0 * (1.0 - a3_s0))
        high = 1.0 / (1.0 + np.exp(-s))
    a3 = np.where(s < s0, low, high)

    b0 = (a3 ** gam) * 10.0 ** (a0 + a1 * wspd)

    b1 = c[15] * wspd * (0.5 + x - np.tanh(4.0 * (x + c[16] + c[17] * wspd)))
    b1 = (c[14] * (1.0 + x) - b1) / (np.exp(0.34 * (wspd - c[18])) + 1.0)

    v0 = c[21] + c[22] * x + c[23] * x2
    d1 = c[24] + c[25] * x + c[26] * x2
    d2 = c[27] + c[28] * x
    v2 = wspd / v0 + 1.0
    with np.errstate(all="ignore"):
        v2_low = a + b * (v2 - 1.0) ** pn
    v2 = np.where(v2 < y0, v2_low, v2)

    b2 = (-d1 + d2 * v2) * np.exp(-v2)

    with np.errstate(all="ignore"):
        sig = b0 * (1.0 + b1 * cosphi + b2 * (2.0 * cosphi ** 2.0 - 1.0)) ** zpow
    return sig


def gmf_cmod5(inc, wspd, phi):
    return _cmod5_core(_C_CMOD5, inc, wspd, phi)


def gmf_cmod5n(inc, wspd, phi):
    return _cmod5_core(_C_CMOD5N, inc, wspd, phi)


def gmf_cmod5n_pr_zhangA(inc, wspd, phi):
    """gmfs_impl.py:93-99, :165-172: sigma0_HH = sigma0_VV / (ar(inc) * wspd**br(inc))."""
    inc = np.asarray(inc, dtype=np.float64)
    wspd = np.asarray(wspd, dtype=np.float64)
    sig = _cmod5_core(_C_CMOD5N, inc, wspd, phi)
    ars2 = np.polynomial.polynomial.polyval(inc, np.array([1.3794, -3.19e-2, 1.4e-3]))
    brs2 = np.polynomial.polynomial.polyval(inc, np.array([-0.1711, 2.6e-3]))
    pr = ars2 * (wspd ** brs2)
    return sig / pr


def gmf_cmod5n_pr_mouche1(inc, wspd, phi):
    """gmfs_impl.py:100-114, :174-199 (Mouche et al. 2005 polarisation ratio)."""
    inc = np.asarray(inc, dtype=np.float64)
    phi = np.asarray(phi, dtype=np.float64)
    sig = _cmod5_core(_C_CMOD5N, inc, wspd, phi)
    p0 = 0.00650704 * np.exp(0.128983 * inc) + 0.992839
    ppi2 = 0.00782194 * np.exp(0.121405 * inc) + 0.992839
    ppi = 0.00598416 * np.exp(0.140952 * inc) + 0.992885
    c0 = (p0 + ppi + 2 * ppi2) / 4
    c1 = (p0 - ppi) / 2
    c2 = (p0 + ppi - 2 * ppi2) / 4
    pr = c0 + c1 * np.cos(np.deg2rad(phi)) + c2 * np.cos(2 * np.deg2rad(phi))
    return sig / pr


_C_IFR2 = (
    0.0, -2.437597, -1.5670307, 0.3708242, -0.040590, 0.404678, 0.188397, -0.027262, 0.064650,
    0.054500, 0.086350, 0.055100, -0.058450, -0.096100, 0.412754, 0.121785, -0.024333, 0.072163,
    -0.062954, 0.015958, -0.069514, -0.062945, 0.035538, 0.023049, 0.074654, -0.014713,
)


def gmf_cmodifr2(inc, wspd, phi):
    """gmfs_impl.py:213-303."""
    C = _C_IFR2
    T = np.asarray(inc, dtype=np.float64)
    wind = np.asarray(wspd, dtype=np.float64)
    ang = np.asarray(phi, dtype=np.float64)

    tetai = (T - 36.0) / 19.0
    xSQ = tetai * tetai
    P1 = tetai
    P2 = (3.0 * xSQ - 1.0) / 2.0
    P3 = (5.0 * xSQ - 3.0) * tetai / 2.0
    ALPH = C[1] + C[2] * P1 + C[3] * P2 + C[4] * P3
    BETA = C[5] + C[6] * P1 + C[7] * P2
    cosi = np.cos(np.deg2rad(ang))
    cos2i = 2.0 * cosi * cosi - 1.0
    tetanor = (2.0 * T - (18.0 + 58.0)) / (58.0 - 18.0)
    vitnor = (2.0 * wind - (25.0 + 3.0)) / (25.0 - 3.0)
    pv0 = 1.0
    pv1 = vitnor
    pv2 = 2 * vitnor * pv1 - pv0
    pv3 = 2 * vitnor * pv2 - pv1
    pt0 = 1.0
    pt1 = tetanor
    pt2 = 2 * tetanor * pt1 - pt0
    b1 = C[8] + C[9] * pv1 + (C[10] + C[11] * pv1) * pt1 + (C[12] + C[13] * pv1) * pt2
    b2 = (
        C[14]
        + C[15] * pt1
        + C[16] * pt2
        + (C[17] + C[18] * pt1 + C[19] * pt2) * pv1
        + (C[20] + C[21] * pt1 + C[22] * pt2) * pv2
        + (C[23] + C[24] * pt1 + C[25] * pt2) * pv3
    )
    b0 = np.power(10.0, (ALPH + BETA * np.sqrt(wind)))
    return b0 * (1.0 + b1 * cosi + np.tanh(b2) * cos2i)


# --- cross-pol (VH) GMFs: two power-law regimes Z1, Z2 blended by two sigmoids -------------
# (a0_Z1, b0_Z1, b1_Z1), (a0_Z2, a1_Z2, a2_Z2, b0_Z2, b1_Z2, b2_Z2), (c0, c1, c2, c3), b0_Z2 scale
_VH_SUM = {
    "gmf_rs2_v2": (
        (6.55519203e-06, 2.49753154e00, -1.35734881e-02),
        (1.47342197e-04, -4.07334797e-06, 3.43593382e-08, 1.10188639e00, 1.40782758e-02, -1.53748743e-04),
        (-0.18675905, 24.48859492, 0.19185442, 25.38275738),
    ),
    "gmf_s1_v2": (
        (2.13755392e-06, 2.47395267e00, -2.85775085e-03),
        (6.54058552e-05, -2.43845137e-06, 2.87698338e-08, 1.14509104e00, 3.41828829e-02, -4.79715441e-04),
        (-0.23257086, 12.39717002, 0.21667263, 12.22862991),
    ),
    "gmf_rcm_noaa": (
        (2.2309436836414871e-12, 8.3374911282878728, -0.033443488982800210),
        (7.7945050373193260e-05, -2.4425748662769216e-06, 2.7625550632547159e-08,
         1.2524896108831316, 0.019203092214131894, -0.00028408046502692580),
        (-0.34498737004629487, 12.558975188752012, 0.12713502524515713, 4.2806865431046752),
    ),
}
_VH_DB = {
    "gmf_s1_v3_ew_rec": (
        (3.5033427638479895e-06, 2.5486758595982275, -0.009042529888607539),
        (4.142689709809047e-05, -1.6620917447744406e-06, 2.4331104610101826e-08,
         1.277314996198736, 0.03813903872809897, -0.0006506765114704733),
        (-0.2522916645939956, 15.3393676653533, 0.24259895576004784, 15.203063214062643),
        1.0,
    ),
    "gmf_rs2_v3": (
        (8.423384272498706e-06, 2.4351127340627374, -0.01450322326682606),
        (0.00014955206131320428, -4.737691852310481e-06, 3.813107432709729e-08,
         1.524883207000445, -0.01322253424944054, 0.00037527120092119504),
        (-0.2222881984904166, 13.118282628673661, 0.21426139278646567, 12.768845054319682),
        1.0,
    ),
    "gmf_rcm_v3": (
        (7.093964676135241e-06, 2.3722948391886542, -0.009516840375089524),
        (6.689451099284358e-05, -1.3956325894252652e-06, 9.227949977841212e-09,
         1.4687699534267797, 0.005735224541037088, -7.164130353316848e-05),
        (-0.2454472887447197, 15.537961353644508, 0.24011368010838255, 15.332883245452303),
        1.0,
    ),
}
_VH_DB["gmf_rcm_v4"] = _VH_DB["gmf_rcm_v3"][:3] + (1.01,)
_VH_DB["gmf_rs2_v4"] = _VH_DB["gmf_rs2_v3"][:3] + (1.01,)


def _vh_regimes(z1, z2, cc, b0_scale, inc, u10):
    inc = np.asarray(inc, dtype=np.float64)
    u10 = np.asarray(u10, dtype=np.float64)
    a_z1 = z1[0]
    b_z1 = z1[1] + z1[2] * inc
    sig_z1 = a_z1 * u10 ** (b_z1)
    a_z2 = z2[0] + z2[1] * inc + z2[2] * inc ** 2
    if b0_scale == 1.0:
        b_z2 = z2[3] + z2[4] * inc + z2[5] * inc ** 2
    else:
        b_z2 = z2[3] * b0_scale + z2[4] * inc + z2[5] * inc ** 2
    sig_z2 = a_z2 * u10 ** (b_z2)
    sigmoid1 = 1 / (1 + np.exp(-cc[0] * (u10 - cc[1])))
    sigmoid2 = 1 / (1 + np.exp(-cc[2] * (u10 - cc[3])))
    return sig_z1, sig_z2, sigmoid1, sigmoid2


def _make_vh_sum(name):
    z1, z2, cc = _VH_SUM[name]

    def f(inc, wspd, phi=None):
        s1, s2, g1, g2 = _vh_regimes(z1, z2, cc, 1.0, inc, wspd)
        return s1 * g1 + s2 * g2

    f.__name__ = name
    return f


def _make_vh_db(name):
    z1, z2, cc, scale = _VH_DB[name]

    def f(inc, wspd, phi=None):
        s1, s2, g1, g2 = _vh_regimes(z1, z2, cc, scale, inc, wspd)
        return 10 ** ((10 * np.log10(s1) * g1 + 10 * np.log10(s2) * g2) / 10)

    f.__name__ = name
    return f


def gmf_dummy(inc, wspd, phi=None):
    """The example GMF of the reference's docs/tests (gmfs.py:45-57, test_xsarsea.py:8-21)."""
    inc = np.asarray(inc, dtype=np.float64)
    wspd = np.asarray(wspd, dtype=np.float64)
    a = 0.00013106836021008122 + -4.530598283705591e-06 * inc + 4.429277425062766e-08 * inc ** 2
    b = 1.3925444179360706 + 0.004157838450541205 * inc + 3.4735809771069953e-05 * inc ** 2
    return a * wspd ** b


# name -> (function, pol, wspd_range, phi_range)   (registration defaults gmfs.py:87-95)
GMFS = {
    "gmf_cmod5": (gmf_cmod5, "VV", [0.2, 50.0], [0.0, 180.0]),
    "gmf_cmod5n": (gmf_cmod5n, "VV", [0.2, 50.0], [0.0, 180.0]),
    "gmf_cmod5n_pr_zhangA": (gmf_cmod5n_pr_zhangA, "HH", [0.2, 50.0], [0.0, 180.0]),
    "gmf_cmod5n_pr_mouche1": (gmf_cmod5n_pr_mouche1, "HH", [0.2, 50.0], [0.0, 180.0]),
    "gmf_cmodifr2": (gmf_cmodifr2, "VV", [0.2, 50.0], [0.0, 180.0]),
}
for _n in _VH_SUM:
    GMFS[_n] = (_make_vh_sum(_n), "VH", [3.0, 80.0], None)
for _n in _VH_DB:
    GMFS[_n] = (_make_vh_db(_n), "VH", [3.0, 80.0], None)
