// Forward GMFs on the device: sigma0 = f(incidence, wind speed[, direction]) for the built-in
// analytic models, elementwise over already-broadcast arrays.  Replaces what numba compiles from the scalar
// Python functions (windspeed/gmfs.py:202-236, `vectorize(target="parallel")`); formulas: published CMOD5 /
// CMOD5.N (Hersbach), Zhang-A and Mouche polarisation ratios, CMOD-IFR2, IFREMER cross-pol fits (the
// coefficient sets are the ones of windspeed/gmfs_impl.py).  float64; transcendental functions are ocml's
// (1-2 ulp), so values agree with the host numpy evaluation to ~1e-14 relative, not bit for bit.
#pragma once
#include <hip/hip_runtime.h>

namespace xsw {

enum GmfId : int {
    GMF_CMOD5 = 0, GMF_CMOD5N = 1, GMF_CMOD5N_ZHANGA = 2, GMF_CMOD5N_MOUCHE1 = 3, GMF_CMODIFR2 = 4,
    GMF_RS2_V2 = 5, GMF_S1_V2 = 6, GMF_RCM_NOAA = 7, GMF_S1_V3_EW_REC = 8, GMF_RS2_V3 = 9, GMF_RCM_V3 = 10,
    GMF_RCM_V4 = 11, GMF_RS2_V4 = 12, GMF_COUNT = 13
};

// (constexpr, not __constant__: with the model a template argument the terms that depend on the coefficients alone -- a, b,
// pow(y0 - 1, pn - 1) -- fold at compile time and nothing is indexed at run time: round 4's one-kernel-for-all form carried the
// table pointer through a switch and spilled, 181 VGPRs + 24 B of scratch)
constexpr double kCmod5_c[29] = {0.0, -0.688, -0.793, 0.338, -0.173, 0.0, 0.004, 0.111, 0.0162, 6.34, 2.57, -2.18, 0.4, -0.6, 0.045, 0.007, 0.33,
                                 0.012, 22.0, 1.95, 3.0, 8.39, -3.44, 1.36, 5.35, 1.99, 0.29, 3.80, 1.53};
constexpr double kCmod5n_c[29] = {0.0, -0.6878, -0.7957, 0.338, -0.1728, 0.0, 0.004, 0.1103, 0.0159, 6.7329, 2.7713, -2.2885, 0.4971, -0.725, 0.045,
                                  0.0066, 0.3222, 0.012, 22.7, 2.0813, 3.0, 8.3659, -3.3428, 1.3236, 6.2437, 2.3893, 0.3249, 4.159, 1.693};

__constant__ double kIfr2[26] = {0.0, -2.437597, -1.5670307, 0.3708242, -0.040590, 0.404678, 0.188397, -0.027262,
                                 0.064650, 0.054500, 0.086350, 0.055100, -0.058450, -0.096100, 0.412754, 0.121785,
                                 -0.024333, 0.072163, -0.062954, 0.015958, -0.069514, -0.062945, 0.035538, 0.023049,
                                 0.074654, -0.014713};

// cross-pol: z1 (a0,b0,b1), z2 (a0,a1,a2,b0,b1,b2), logistic (k0,k1,k2,k3), blend (0 linear / 1 dB), scale of z2.b0
__constant__ double kVh[8][15] = {
    {6.55519203e-06, 2.49753154e00, -1.35734881e-02, 1.47342197e-04, -4.07334797e-06, 3.43593382e-08, 1.10188639e00,
     1.40782758e-02, -1.53748743e-04, -0.18675905, 24.48859492, 0.19185442, 25.38275738, 0, 1.0},
    {2.13755392e-06, 2.47395267e00, -2.85775085e-03, 6.54058552e-05, -2.43845137e-06, 2.87698338e-08, 1.14509104e00,
     3.41828829e-02, -4.79715441e-04, -0.23257086, 12.39717002, 0.21667263, 12.22862991, 0, 1.0},
    {2.2309436836414871e-12, 8.3374911282878728, -0.033443488982800210, 7.7945050373193260e-05, -2.4425748662769216e-06,
     2.7625550632547159e-08, 1.2524896108831316, 0.019203092214131894, -0.00028408046502692580, -0.34498737004629487,
     12.558975188752012, 0.12713502524515713, 4.2806865431046752, 0, 1.0},
    {3.5033427638479895e-06, 2.5486758595982275, -0.009042529888607539, 4.142689709809047e-05, -1.6620917447744406e-06,
     2.4331104610101826e-08, 1.277314996198736, 0.03813903872809897, -0.0006506765114704733, -0.2522916645939956,
     15.3393676653533, 0.24259895576004784, 15.203063214062643, 1, 1.0},
    {8.423384272498706e-06, 2.4351127340627374, -0.01450322326682606, 0.00014955206131320428, -4.737691852310481e-06,
     3.813107432709729e-08, 1.524883207000445, -0.01322253424944054, 0.00037527120092119504, -0.2222881984904166,
     13.118282628673661, 0.21426139278646567, 12.768845054319682, 1, 1.0},
    {7.093964676135241e-06, 2.3722948391886542, -0.009516840375089524, 6.689451099284358e-05, -1.3956325894252652e-06,
     9.227949977841212e-09, 1.4687699534267797, 0.005735224541037088, -7.164130353316848e-05, -0.2454472887447197,
     15.537961353644508, 0.24011368010838255, 15.332883245452303, 1, 1.0},
    {7.093964676135241e-06, 2.3722948391886542, -0.009516840375089524, 6.689451099284358e-05, -1.3956325894252652e-06,
     9.227949977841212e-09, 1.4687699534267797, 0.005735224541037088, -7.164130353316848e-05, -0.2454472887447197,
     15.537961353644508, 0.24011368010838255, 15.332883245452303, 1, 1.01},
    {8.423384272498706e-06, 2.4351127340627374, -0.01450322326682606, 0.00014955206131320428, -4.737691852310481e-06,
     3.813107432709729e-08, 1.524883207000445, -0.01322253424944054, 0.00037527120092119504, -0.2222881984904166,
     13.118282628673661, 0.21426139278646567, 12.768845054319682, 1, 1.01}};

template <bool NEUTRAL>
__device__ __forceinline__ double gmf_cmod5_family(double inc, double v, double phi)
{
    constexpr const double *c = NEUTRAL ? kCmod5n_c : kCmod5_c;
    const double cosphi = cos(phi * (M_PI / 180.0));
    const double x = (inc - 40.0) / 25.0, x2 = x * x;
    const double y0 = c[19], pn = c[20];
    const double a = y0 - (y0 - 1.0) / pn;
    const double b = 1.0 / (pn * pow(y0 - 1.0, pn - 1.0));
    const double a0 = c[1] + c[2] * x + c[3] * x2 + c[4] * x * x2;
    const double a1 = c[5] + c[6] * x, a2 = c[7] + c[8] * x;
    const double gam = c[9] + c[10] * x + c[11] * x2;
    const double s0 = c[12] + c[13] * x;
    const double s = a2 * v;
    const double sig0 = 1.0 / (1.0 + exp(-s0));
    const double a3 = (s < s0) ? sig0 * pow(s / s0, s0 * (1.0 - sig0)) : 1.0 / (1.0 + exp(-s));
    const double b0 = pow(a3, gam) * pow(10.0, a0 + a1 * v);
    double b1 = c[15] * v * (0.5 + x - tanh(4.0 * (x + c[16] + c[17] * v)));
    b1 = (c[14] * (1.0 + x) - b1) / (exp(0.34 * (v - c[18])) + 1.0);
    const double v0 = c[21] + c[22] * x + c[23] * x2;
    const double d1 = c[24] + c[25] * x + c[26] * x2;
    const double d2 = c[27] + c[28] * x;
    double v2 = v / v0 + 1.0;
    if (v2 < y0) v2 = a + b * pow(v2 - 1.0, pn);
    const double b2 = (-d1 + d2 * v2) * exp(-v2);
    return b0 * pow(1.0 + b1 * cosphi + b2 * (2.0 * cosphi * cosphi - 1.0), 1.6);
}

__device__ inline double gmf_ifr2(double T, double wind, double ang)
{
    const double *C = kIfr2;
    const double t = (T - 36.0) / 19.0, t2 = t * t;
    const double P1 = t, P2 = (3.0 * t2 - 1.0) / 2.0, P3 = (5.0 * t2 - 3.0) * t / 2.0;
    const double ALPH = C[1] + C[2] * P1 + C[3] * P2 + C[4] * P3;
    const double BETA = C[5] + C[6] * P1 + C[7] * P2;
    const double cosi = cos(ang * (M_PI / 180.0)), cos2i = 2.0 * cosi * cosi - 1.0;
    const double tn = (2.0 * T - 76.0) / 40.0, vn = (2.0 * wind - 28.0) / 22.0;
    const double pv1 = vn, pv2 = 2 * vn * pv1 - 1.0, pv3 = 2 * vn * pv2 - pv1;
    const double pt1 = tn, pt2 = 2 * tn * pt1 - 1.0;
    const double b1 = C[8] + C[9] * pv1 + (C[10] + C[11] * pv1) * pt1 + (C[12] + C[13] * pv1) * pt2;
    const double b2 = C[14] + C[15] * pt1 + C[16] * pt2 + (C[17] + C[18] * pt1 + C[19] * pt2) * pv1 +
                      (C[20] + C[21] * pt1 + C[22] * pt2) * pv2 + (C[23] + C[24] * pt1 + C[25] * pt2) * pv3;
    return pow(10.0, ALPH + BETA * sqrt(wind)) * (1.0 + b1 * cosi + tanh(b2) * cos2i);
}

__device__ inline double gmf_vh(const double *p, double inc, double u)
{
    const double s1 = p[0] * pow(u, p[1] + p[2] * inc);
    const double a2 = p[3] + p[4] * inc + p[5] * inc * inc;
    const double s2 = a2 * pow(u, p[6] * p[14] + p[7] * inc + p[8] * inc * inc);
    const double w1 = 1.0 / (1.0 + exp(-p[9] * (u - p[10])));
    const double w2 = 1.0 / (1.0 + exp(-p[11] * (u - p[12])));
    if (p[13] == 0.0) return s1 * w1 + s2 * w2;
    return pow(10.0, (10.0 * log10(s1) * w1 + 10.0 * log10(s2) * w2) / 10.0);
}

// the model as a template argument (GMF_RS2_V2 stands for the whole cross-pol family: one code path, coefficients by `id`)
template <int M>
__device__ __forceinline__ double gmf_eval(int id, double inc, double v, double phi)
{
    if (M == GMF_CMOD5) return gmf_cmod5_family<false>(inc, v, phi);
    if (M == GMF_CMOD5N) return gmf_cmod5_family<true>(inc, v, phi);
    if (M == GMF_CMOD5N_ZHANGA) {
        const double ar = 1.3794 + inc * (-3.19e-2 + inc * 1.4e-3), br = -0.1711 + inc * 2.6e-3;
        return gmf_cmod5_family<true>(inc, v, phi) / (ar * pow(v, br));
    }
    if (M == GMF_CMOD5N_MOUCHE1) {
        const double p0 = 0.00650704 * exp(0.128983 * inc) + 0.992839, ph = 0.00782194 * exp(0.121405 * inc) + 0.992839;
        const double pp = 0.00598416 * exp(0.140952 * inc) + 0.992885, r = phi * (M_PI / 180.0);
        const double pr = (p0 + pp + 2 * ph) / 4 + (p0 - pp) / 2 * cos(r) + (p0 + pp - 2 * ph) / 4 * cos(2 * r);
        return gmf_cmod5_family<true>(inc, v, phi) / pr;
    }
    if (M == GMF_CMODIFR2) return gmf_ifr2(inc, v, phi);
    return gmf_vh(kVh[id - GMF_RS2_V2], inc, v);
}

// out[i] = gmf(inc[i], wspd[i], phi[i]); phi may be NULL for the cross-pol models
template <int M>
__global__ __launch_bounds__(256) void k_gmf_eval(int id, long long n, const double *__restrict__ inc,
                                                  const double *__restrict__ wspd, const double *__restrict__ phi,
                                                  double *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = gmf_eval<M>(id, inc[i], wspd[i], phi ? phi[i] : 0.0);
}

// host side: the instantiation of model `id` (the cross-pol models share one)
#define XSW_GMF_DISPATCH(id, CALL)                                   \
    switch (id) {                                                    \
    case GMF_CMOD5: { constexpr int M = GMF_CMOD5; CALL; } break;     \
    case GMF_CMOD5N: { constexpr int M = GMF_CMOD5N; CALL; } break;   \
    case GMF_CMOD5N_ZHANGA: { constexpr int M = GMF_CMOD5N_ZHANGA; CALL; } break;   \
    case GMF_CMOD5N_MOUCHE1: { constexpr int M = GMF_CMOD5N_MOUCHE1; CALL; } break; \
    case GMF_CMODIFR2: { constexpr int M = GMF_CMODIFR2; CALL; } break;             \
    default: { constexpr int M = GMF_RS2_V2; CALL; } break;          \
    }

}  // namespace xsw
