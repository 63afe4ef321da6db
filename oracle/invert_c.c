/* CPU oracle (TEST INFRASTRUCTURE ONLY): plain-C restatement of the reference's per-pixel
 * inversion kernel, /root/reference/src/xsarsea/windspeed/windspeed.py:183-282
 * (`__invert_from_model_1d`), with the row-parallel execution of its numba wrapper
 * (`guvectorize(..., target="parallel")`, windspeed.py:306-323) rendered as an OpenMP loop.
 *
 * Faithful to the reference on purpose (it is also the reported CPU baseline, kind "port"):
 *   - strict IEEE float64, no contraction, no fast-math (windspeed.py:320 passes a dict for
 *     `fastmath`, i.e. no fast-math flags); build with -ffp-contract=off;
 *   - the co-pol LUT is addressed in the reference's layout (wspd, phi, incidence) C-contiguous
 *     (windspeed.py:145-147), so one incidence slice is a strided gather exactly as at :213;
 *   - the incidence bin is a full linear scan argmin(|inc_dim - inc|), first minimum (:212);
 *   - the cost is ((cx-a)/2)^2 + ((cy-b)/2)^2 + ((lut-s)/dsig)^2 with a true division (:220-225);
 *   - argmin is numpy's: first minimum, but a NaN anywhere wins (first NaN) (:228).
 * Used only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    /* co-pol (n_wspd == 0 when there is no co-pol model, windspeed.py:157-164) */
    const double *co_lut;       /* [n_wspd][n_phi][n_inc] dB */
    const double *wspd_dim, *phi_dim, *inc_dim;
    const double *lut_antenna;  /* [n_wspd][n_phi]  wspd*cos(radians(phi))  (:167) */
    const double *lut_azi;      /* [n_wspd][n_phi]  wspd*sin(radians(phi))  (:168) */
    int32_t n_wspd, n_phi, n_inc, phi_180;
    double dsig_co;
    /* cross-pol (n_wspd_cr == 0 when unused, :177-181) */
    const double *cr_lut;       /* [n_wspd_cr][n_inc_cr] dB */
    const double *wspd_cr, *inc_cr_dim;
    int32_t n_wspd_cr, n_inc_cr;
    /* element strides: reference layout = (n_inc, 1) / (n_inc_cr, 1); the tests' fast variant passes
     * an incidence-major copy with (1, n_wspd*n_phi) / (1, n_wspd_cr): same arithmetic, same results,
     * contiguous slice. */
    int64_t co_stride_cand, co_stride_inc, cr_stride_cand, cr_stride_inc;
} oracle_luts;

/* numpy argmin over |dim - x| : first minimum; NaN would win but x is not NaN here. */
static int nearest_index(const double *dim, int n, double x)
{
    int best = 0;
    double bv = fabs(dim[0] - x);
    if (isnan(bv)) return 0;
    for (int k = 1; k < n; ++k) {
        double v = fabs(dim[k] - x);
        if (isnan(v)) return k;
        if (v < bv) { bv = v; best = k; }
    }
    return best;
}

/* numpy's complex128 true-divide loop (Smith's algorithm), then np.angle = atan2(im, re). */
static double angle_of_quotient(double ar, double ai, double br, double bi)
{
    double qr, qi;
    double abr = fabs(br), abi = fabs(bi);
    if (abr >= abi) {
        if (abr == 0 && abi == 0) { qr = ar / abr; qi = ai / abi; }
        else {
            double rat = bi / br, scl = 1.0 / (br + bi * rat);
            qr = (ar + ai * rat) * scl;
            qi = (ai - ar * rat) * scl;
        }
    } else {
        double rat = br / bi, scl = 1.0 / (bi + br * rat);
        qr = (ar * rat + ai) * scl;
        qi = (ai * rat - ar) * scl;
    }
    return atan2(qi, qr);
}

static void one_pixel(const oracle_luts *L, double inc, double s_co, double s_cr, double dsig_cr,
                      double a_re, double a_im, double *out_co, double *out_cr, int64_t *idx)
{
    const double nan = NAN;
    if (idx) { idx[0] = idx[1] = idx[2] = -1; }
    if (isnan(inc)) { /* :198-201 */
        out_co[0] = nan; out_co[1] = 0.0; out_cr[0] = nan; out_cr[1] = 0.0;
        return;
    }
    /* :204-207  np.isnan(np.abs(z)) for complex z = isnan(hypot(re, im)) */
    if (!isnan(fabs(s_co)) && isnan(hypot(a_re, a_im))) {
        out_co[0] = nan; out_co[1] = 0.0; out_cr[0] = nan; out_cr[1] = 0.0;
        return;
    }
    double co_re, co_im;
    if (!isnan(s_co)) { /* :209-247 */
        int i_inc = nearest_index(L->inc_dim, L->n_inc, inc);
        double m_ant = a_re, m_azi = a_im;
        if (L->phi_180) m_azi = fabs(m_azi);
        int64_t best = 0;
        double bestJ = 0.0;
        int have = 0, nan_hit = 0;
        const int64_t ncand = (int64_t)L->n_wspd * L->n_phi;
        for (int64_t c = 0; c < ncand && !nan_hit; ++c) {
            double t1 = (L->lut_antenna[c] - m_ant) / 2.0;
            double t2 = (L->lut_azi[c] - m_azi) / 2.0;
            double jw = t1 * t1 + t2 * t2;
            double d = (L->co_lut[c * L->co_stride_cand + i_inc * L->co_stride_inc] - s_co) / L->dsig_co;
            double J = jw + d * d;
            if (isnan(J)) { best = c; nan_hit = 1; break; }
            if (!have || J < bestJ) { bestJ = J; best = c; have = 1; }
        }
        int iw = (int)(best / L->n_phi), ip = (int)(best % L->n_phi);
        if (idx) { idx[0] = iw; idx[1] = ip; }
        double w = L->wspd_dim[iw], phi = L->phi_dim[ip];
        const double d2r = M_PI / 180.0; /* np.deg2rad(x) = x * (pi/180) */
        double r1 = phi * d2r;
        double s1r = w * cos(r1), s1i = w * sin(r1) + 0.0 * cos(r1);
        if (L->phi_180) { /* :234-242 */
            double r2 = (-phi) * d2r;
            double s2r = w * cos(r2), s2i = w * sin(r2) + 0.0 * cos(r2);
            double d1 = angle_of_quotient(a_re, a_im, s1r, s1i);
            double d2 = angle_of_quotient(a_re, a_im, s2r, s2i);
            if (fabs(d1) <= fabs(d2)) { co_re = s1r; co_im = s1i; }
            else { co_re = s2r; co_im = s2i; }
        } else {
            co_re = s1r; co_im = s1i;
        }
    } else {
        co_re = nan; co_im = nan; /* np.nan * 1j = (nan + nanj) */
    }

    double cr_re, cr_im;
    if (!isnan(s_cr) && !isnan(dsig_cr)) { /* :252-276 */
        int i_inc = nearest_index(L->inc_cr_dim, L->n_inc_cr, inc);
        double aco = hypot(co_re, co_im);
        int have_co = !isnan(aco);
        int best = 0, have = 0;
        double bestJ = 0.0;
        for (int k = 0; k < L->n_wspd_cr; ++k) {
            double d = (L->cr_lut[k * L->cr_stride_cand + i_inc * L->cr_stride_inc] - s_cr) / dsig_cr;
            double J = d * d;
            if (have_co) {
                double t = (L->wspd_cr[k] - aco) / 2.0;
                J = J + t * t; /* J_cr = Jsig_cr + Jwind_cr  (:261) */
            }
            if (isnan(J)) { best = k; break; }
            if (!have || J < bestJ) { bestJ = J; best = k; have = 1; }
        }
        if (idx) idx[2] = best;
        double wd = L->wspd_cr[best];
        double ph = have_co ? atan2(co_im, co_re) : 0.0;
        cr_re = wd * cos(ph);
        cr_im = wd * sin(ph) + 0.0 * cos(ph);
    } else {
        cr_re = nan; cr_im = nan;
    }
    out_co[0] = co_re; out_co[1] = co_im;
    out_cr[0] = cr_re; out_cr[1] = cr_im;
}

/* Inputs are the gufunc's (float64 x4, complex128 interleaved); outputs complex128 interleaved.
 * `idx` (nullable) receives (i_wspd, i_phi, i_wspd_cr) per pixel, -1 where no search ran. */
int oracle_invert(const oracle_luts *L, int64_t n, const double *inc, const double *s_co_db,
                  const double *s_cr_db, const double *dsig_cr, const double *anc,
                  double *out_co, double *out_cr, int64_t *idx, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < n; ++i)
        one_pixel(L, inc[i], s_co_db[i], s_cr_db[i], dsig_cr[i], anc[2 * i], anc[2 * i + 1],
                  out_co + 2 * i, out_cr + 2 * i, idx ? idx + 3 * i : 0);
    return 0;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
