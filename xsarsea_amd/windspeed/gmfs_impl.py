"""Built-in analytic GMFs, array-wise (host side; they fill the raw LUT and serve sigma0_detrend).

Same model set and registration names as the reference's `xsarsea/windspeed/gmfs_impl.py`
(CMOD5, CMOD5.N, its two HH polarisation-ratio variants, CMOD-IFR2 and eight Sentinel-1 /
Radarsat-2 / RCM cross-pol GMFs).  The formulas are the published ones (Hersbach 2007/2010 for
CMOD5/5.N, Quilfen 1998 for IFR2, Mouche 2005 / Zhang 2011 polarisation ratios, IFREMER VH fits);
each family is one evaluator driven by a coefficient record.  Every function accepts
broadcastable float64 arrays (or scalars) and returns linear sigma0.
"""
import numpy as np

from .gmfs import GmfModel

_D2R = np.pi / 180.0


def _f64(*xs):
    return tuple(np.asarray(x, dtype=np.float64) for x in xs)


def _need_phi(phi):
    if phi is None:
        raise TypeError("this GMF depends on the wind direction: phi is required")


# ------------------------------------------------------------------------------------------- CMOD5
# c[1..28] of the CMOD5 family (index 0 unused so that the indices are the published ones)
_CMOD5 = np.array([0.0, -0.688, -0.793, 0.338, -0.173, 0.0, 0.004, 0.111, 0.0162, 6.34, 2.57, -2.18, 0.4, -0.6, 0.045,
                   0.007, 0.33, 0.012, 22.0, 1.95, 3.0, 8.39, -3.44, 1.36, 5.35, 1.99, 0.29, 3.80, 1.53])
_CMOD5N = np.array([0.0, -0.6878, -0.7957, 0.338, -0.1728, 0.0, 0.004, 0.1103, 0.0159, 6.7329, 2.7713, -2.2885,
                    0.4971, -0.725, 0.045, 0.0066, 0.3222, 0.012, 22.7, 2.0813, 3.0, 8.3659, -3.3428, 1.3236, 6.2437,
                    2.3893, 0.3249, 4.159, 1.693])


def _cmod5_sigma0(c, inc, wspd, phi):
    """sigma0 = B0 * (1 + B1 cos(phi) + B2 cos(2 phi))**1.6 with the CMOD5 B-terms."""
    _need_phi(phi)
    inc, v, phi = _f64(inc, wspd, phi)
    cosphi = np.cos(np.deg2rad(phi))
    x = (inc - 40.0) / 25.0
    x2 = x ** 2.0
    y0, pn = c[19], c[20]
    a = y0 - (y0 - 1.0) / pn
    b = 1.0 / (pn * (y0 - 1.0) ** (pn - 1.0))

    # B0
    a0 = c[1] + c[2] * x + c[3] * x2 + c[4] * x * x2
    a1 = c[5] + c[6] * x
    a2 = c[7] + c[8] * x
    gam = c[9] + c[10] * x + c[11] * x2
    s0 = c[12] + c[13] * x
    s = a2 * v
    sig_s0 = 1.0 / (1.0 + np.exp(-s0))
    with np.errstate(all="ignore"):
        below = sig_s0 * (s / s0) ** (s0 * (1.0 - sig_s0))
        above = 1.0 / (1.0 + np.exp(-s))
        a3 = np.where(s < s0, below, above)
        b0 = (a3 ** gam) * 10.0 ** (a0 + a1 * v)

        # B1
        b1 = c[15] * v * (0.5 + x - np.tanh(4.0 * (x + c[16] + c[17] * v)))
        b1 = (c[14] * (1.0 + x) - b1) / (np.exp(0.34 * (v - c[18])) + 1.0)

        # B2
        v0 = c[21] + c[22] * x + c[23] * x2
        d1 = c[24] + c[25] * x + c[26] * x2
        d2 = c[27] + c[28] * x
        v2 = v / v0 + 1.0
        v2 = np.where(v2 < y0, a + b * (v2 - 1.0) ** pn, v2)
        b2 = (-d1 + d2 * v2) * np.exp(-v2)

        return b0 * (1.0 + b1 * cosphi + b2 * (2.0 * cosphi ** 2.0 - 1.0)) ** 1.6


def _pr_zhang(inc, wspd):
    """Zhang et al. (2011) model A polarisation ratio PR(inc, wspd)."""
    inc, wspd = _f64(inc, wspd)
    ar = np.polynomial.polynomial.polyval(inc, np.array([1.3794, -3.19e-2, 1.4e-3]))
    br = np.polynomial.polynomial.polyval(inc, np.array([-0.1711, 2.6e-3]))
    return ar * (wspd ** br)


def _pr_mouche(inc, phi):
    """Mouche et al. (2005) polarisation ratio PR(inc, phi)."""
    inc, phi = _f64(inc, phi)
    p0 = 0.00650704 * np.exp(0.128983 * inc) + 0.992839
    phalf = 0.00782194 * np.exp(0.121405 * inc) + 0.992839
    ppi = 0.00598416 * np.exp(0.140952 * inc) + 0.992885
    c0 = (p0 + ppi + 2 * phalf) / 4
    c1 = (p0 - ppi) / 2
    c2 = (p0 + ppi - 2 * phalf) / 4
    return c0 + c1 * np.cos(np.deg2rad(phi)) + c2 * np.cos(2 * np.deg2rad(phi))


@GmfModel.register("gmf_cmod5", wspd_range=[0.2, 50.0], pol="VV", units="linear", defer=False)
def gmf_cmod5(inc, wspd, phi):
    return _cmod5_sigma0(_CMOD5, inc, wspd, phi)


@GmfModel.register("gmf_cmod5n", wspd_range=[0.2, 50.0], pol="VV", units="linear", defer=False)
def gmf_cmod5n(inc, wspd, phi):
    return _cmod5_sigma0(_CMOD5N, inc, wspd, phi)


@GmfModel.register("gmf_cmod5n_pr_zhangA", wspd_range=[0.2, 50.0], pol="HH", units="linear", defer=False)
def gmf_cmod5n_pr_zhangA(inc, wspd, phi):
    return _cmod5_sigma0(_CMOD5N, inc, wspd, phi) / _pr_zhang(inc, wspd)


@GmfModel.register("gmf_cmod5n_pr_mouche1", wspd_range=[0.2, 50.0], pol="HH", units="linear", defer=False)
def gmf_cmod5n_pr_mouche1(inc, wspd, phi):
    return _cmod5_sigma0(_CMOD5N, inc, wspd, phi) / _pr_mouche(inc, phi)


# ------------------------------------------------------------------------------------------- CMOD-IFR2
_IFR2 = np.array([0.0, -2.437597, -1.5670307, 0.3708242, -0.040590, 0.404678, 0.188397, -0.027262, 0.064650, 0.054500,
                  0.086350, 0.055100, -0.058450, -0.096100, 0.412754, 0.121785, -0.024333, 0.072163, -0.062954,
                  0.015958, -0.069514, -0.062945, 0.035538, 0.023049, 0.074654, -0.014713])


@GmfModel.register(wspd_range=[0.2, 50.0], pol="VV", units="linear", defer=False)
def gmf_cmodifr2(inc_angle, wind_speed, wind_dir):
    _need_phi(wind_dir)
    c = _IFR2
    theta, wind, ang = _f64(inc_angle, wind_speed, wind_dir)
    t = (theta - 36.0) / 19.0
    t2 = t * t
    leg1, leg2, leg3 = t, (3.0 * t2 - 1.0) / 2.0, (5.0 * t2 - 3.0) * t / 2.0
    alph = c[1] + c[2] * leg1 + c[3] * leg2 + c[4] * leg3
    beta = c[5] + c[6] * leg1 + c[7] * leg2
    cosi = np.cos(np.deg2rad(ang))
    cos2i = 2.0 * cosi * cosi - 1.0
    tn = (2.0 * theta - (18.0 + 58.0)) / (58.0 - 18.0)
    vn = (2.0 * wind - (25.0 + 3.0)) / (25.0 - 3.0)
    pv1 = vn
    pv2 = 2 * vn * pv1 - 1.0
    pv3 = 2 * vn * pv2 - pv1
    pt1 = tn
    pt2 = 2 * tn * pt1 - 1.0
    b1 = c[8] + c[9] * pv1 + (c[10] + c[11] * pv1) * pt1 + (c[12] + c[13] * pv1) * pt2
    b2 = (c[14] + c[15] * pt1 + c[16] * pt2 + (c[17] + c[18] * pt1 + c[19] * pt2) * pv1
          + (c[20] + c[21] * pt1 + c[22] * pt2) * pv2 + (c[23] + c[24] * pt1 + c[25] * pt2) * pv3)
    with np.errstate(all="ignore"):
        b0 = np.power(10.0, (alph + beta * np.sqrt(wind)))
    return b0 * (1.0 + b1 * cosi + np.tanh(b2) * cos2i)


# ------------------------------------------------------------------------------------------- cross-pol
class _VH:
    """sigma0_VH(inc, u) = blend of two power laws a(inc) * u**b(inc) by two logistic weights."""

    def __init__(self, z1, z2, logistic, blend, b0_scale=None):
        self.z1, self.z2, self.logistic, self.blend, self.b0_scale = z1, z2, logistic, blend, b0_scale

    def __call__(self, incidence, speed, phi=None):
        inc, u = _f64(incidence, speed)
        a0, b0, b1 = self.z1
        with np.errstate(all="ignore"):
            s1 = a0 * u ** (b0 + b1 * inc)
            p = self.z2
            a_z2 = p[0] + p[1] * inc + p[2] * inc ** 2
            lead = p[3] if self.b0_scale is None else p[3] * self.b0_scale
            s2 = a_z2 * u ** (lead + p[4] * inc + p[5] * inc ** 2)
            k0, k1, k2, k3 = self.logistic
            w1 = 1 / (1 + np.exp(-k0 * (u - k1)))
            w2 = 1 / (1 + np.exp(-k2 * (u - k3)))
            if self.blend == "linear":
                return s1 * w1 + s2 * w2
            return 10 ** ((10 * np.log10(s1) * w1 + 10 * np.log10(s2) * w2) / 10)  # blend of the dB values


_RS2_V3 = ((8.423384272498706e-06, 2.4351127340627374, -0.01450322326682606),
           (0.00014955206131320428, -4.737691852310481e-06, 3.813107432709729e-08, 1.524883207000445,
            -0.01322253424944054, 0.00037527120092119504),
           (-0.2222881984904166, 13.118282628673661, 0.21426139278646567, 12.768845054319682))
_RCM_V3 = ((7.093964676135241e-06, 2.3722948391886542, -0.009516840375089524),
           (6.689451099284358e-05, -1.3956325894252652e-06, 9.227949977841212e-09, 1.4687699534267797,
            0.005735224541037088, -7.164130353316848e-05),
           (-0.2454472887447197, 15.537961353644508, 0.24011368010838255, 15.332883245452303))

_VH_MODELS = {
    "gmf_rs2_v2": _VH((6.55519203e-06, 2.49753154e00, -1.35734881e-02),
                      (1.47342197e-04, -4.07334797e-06, 3.43593382e-08, 1.10188639e00, 1.40782758e-02, -1.53748743e-04),
                      (-0.18675905, 24.48859492, 0.19185442, 25.38275738), "linear"),
    "gmf_s1_v2": _VH((2.13755392e-06, 2.47395267e00, -2.85775085e-03),
                     (6.54058552e-05, -2.43845137e-06, 2.87698338e-08, 1.14509104e00, 3.41828829e-02, -4.79715441e-04),
                     (-0.23257086, 12.39717002, 0.21667263, 12.22862991), "linear"),
    "gmf_rcm_noaa": _VH((2.2309436836414871e-12, 8.3374911282878728, -0.033443488982800210),
                        (7.7945050373193260e-05, -2.4425748662769216e-06, 2.7625550632547159e-08, 1.2524896108831316,
                         0.019203092214131894, -0.00028408046502692580),
                        (-0.34498737004629487, 12.558975188752012, 0.12713502524515713, 4.2806865431046752), "linear"),
    "gmf_s1_v3_ew_rec": _VH((3.5033427638479895e-06, 2.5486758595982275, -0.009042529888607539),
                            (4.142689709809047e-05, -1.6620917447744406e-06, 2.4331104610101826e-08, 1.277314996198736,
                             0.03813903872809897, -0.0006506765114704733),
                            (-0.2522916645939956, 15.3393676653533, 0.24259895576004784, 15.203063214062643), "dB"),
    "gmf_rs2_v3": _VH(*_RS2_V3, "dB"),
    "gmf_rcm_v3": _VH(*_RCM_V3, "dB"),
    "gmf_rcm_v4": _VH(*_RCM_V3, "dB", b0_scale=1.01),
    "gmf_rs2_v4": _VH(*_RS2_V3, "dB", b0_scale=1.01),
}


def _register_vh(name, model):
    def f(incidence, speed, phi=None):
        return model(incidence, speed, phi)

    f.__name__ = name
    f.__doc__ = f"{name}: cross-pol sigma0(incidence [deg], speed [m/s]) in linear units."
    GmfModel.register(name, wspd_range=[3.0, 80.0], pol="VH", units="linear", defer=False)(f)
    return f


for _name, _model in _VH_MODELS.items():
    globals()[_name] = _register_vh(_name, _model)


# the models above have a device implementation (csrc/xsw_gmf.hpp); a user model that re-registers one of these
# names with another function does not inherit the flag
for _name in ("gmf_cmod5", "gmf_cmod5n", "gmf_cmod5n_pr_zhangA", "gmf_cmod5n_pr_mouche1", "gmf_cmodifr2", *_VH_MODELS):
    GmfModel._registry[_name]._builtin = True
