// Cross-pol noise flattening on the device (reference: windspeed/utils.py:94-163, `nesz_flattening`): the full-raster
// pass that precedes the dual-pol inversion in the L1 workflow.
//
//   noise_mean[s] = nanmean(noise[:, s])           inc_row[s] = nanmean(inc[:, s])                     (:119-121, :163)
//   per line l:  filled = where(isnan(noise[l]), noise_mean, noise[l]);  y = 10 log10(filled);  ok = isfinite(y)
//                (slope, icpt) = degree-1 least squares of y[ok] against inc_row[ok]                    (:133-149)
//                out[l][s] = 10 ** ((inc_row[s] * slope + icpt - 1) / 10)   for EVERY s                (:152)
//
// Three HBM-bound streaming passes (float32 rasters: 8 B read in the first, 4 B read in the second, 8 B written in the third,
// per pixel) and two small kernels:
//   k_nesz_colsum   partial column sums/counts over blocks of lines (V columns per thread, coalesced rows)
//   k_nesz_colmean  finishes the column means in line-block order (deterministic); k_nesz_center: the centring abscissa x0
//   k_nesz_fit      XSW_NESZ_LINES lines per workgroup: five float64 moments about x0 (n, Sx, Sy, Sxx, Sxy), closed-form fit
//   k_nesz_eval     the raster from the column abscissae and the lines' coefficients (write only).
// Sums are float64 whatever the raster dtype (numpy accumulates float32 rasters in float32); the fit is the closed
// form of the normal equations about x0 instead of numpy's SVD: results agree with the host route to ~1e-13 relative
// for float64 rasters (tests state 1e-10), ~1e-6 for float32 rasters (numpy's float32 log10 and float32 column sums).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xsw {

struct NeszPartial {  // per (line block, column)
    double sum_n, sum_i;
    int cnt_n, cnt_i;
};

// V columns per thread (vector loads when the lines are aligned to them; otherwise -- and for the last, partial group of
// columns -- element by element).  grid.x tiles the column groups, grid.y the line blocks.  The library launches V = 1: wider
// loads measured slower at 20000^2 float32 (V = 1: 0.59 ms, 2: 0.64 ms, 4: 0.66 ms in the same run).
template <typename T, int V>
__global__ __launch_bounds__(256) void k_nesz_colsum(const T *__restrict__ noise, const T *__restrict__ inc,
                                                     NeszPartial *__restrict__ part, long long lines, long long samples,
                                                     long long lines_per_block)
{
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const long long s = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (s >= samples) return;
    const long long l0 = (long long)blockIdx.y * lines_per_block;
    const long long l1 = l0 + lines_per_block < lines ? l0 + lines_per_block : lines;
    double sn[V], si[V];
    int cn[V], ci[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { sn[k] = si[k] = 0.0; cn[k] = ci[k] = 0; }
    auto add = [&](int k, T a, T b) {
        const double x = (double)a, y = (double)b;
        if (x == x) { sn[k] += x; ++cn[k]; }
        if (y == y) { si[k] += y; ++ci[k]; }
    };
    const T *pn = noise + l0 * samples + s, *pi = inc + l0 * samples + s;
    const bool vec = s + V <= samples && (samples % V) == 0 && (((size_t)noise | (size_t)inc) & (sizeof(vec_t) - 1)) == 0;
    if (vec) {
        long long l = l0;
        for (; l + 4 <= l1; l += 4) {  // four lines in flight per lane
            vec_t a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = *(const vec_t *)(pn + u * samples); b[u] = *(const vec_t *)(pi + u * samples); }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < V; ++k) add(k, a[u][k], b[u][k]);
            pn += 4 * samples; pi += 4 * samples;
        }
        for (; l < l1; ++l) {
            const vec_t a = *(const vec_t *)pn, b = *(const vec_t *)pi;
#pragma unroll
            for (int k = 0; k < V; ++k) add(k, a[k], b[k]);
            pn += samples; pi += samples;
        }
    } else {
        for (long long l = l0; l < l1; ++l) {
#pragma unroll
            for (int k = 0; k < V; ++k)
                if (s + k < samples) add(k, pn[k], pi[k]);
            pn += samples; pi += samples;
        }
    }
#pragma unroll
    for (int k = 0; k < V; ++k)
        if (s + k < samples) {
            NeszPartial p;
            p.sum_n = sn[k]; p.sum_i = si[k]; p.cnt_n = cn[k]; p.cnt_i = ci[k];
            part[(long long)blockIdx.y * samples + s + k] = p;
        }
}

// col[0][s] = noise_mean, col[1][s] = inc_row (NaN for a column without a valid sample, as np.nanmean)
__global__ __launch_bounds__(256) void k_nesz_colmean(const NeszPartial *__restrict__ part, double *__restrict__ col,
                                                      long long samples, int nblocks)
{
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= samples) return;
    double sn = 0.0, si = 0.0;
    long long cn = 0, ci = 0;
    for (int b = 0; b < nblocks; ++b) {
        const NeszPartial p = part[(long long)b * samples + s];
        sn += p.sum_n; si += p.sum_i; cn += p.cnt_n; ci += p.cnt_i;
    }
    const double nan = __builtin_nan("");
    col[s] = cn ? sn / (double)cn : nan;
    col[samples + s] = ci ? si / (double)ci : nan;
}

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// x0 = mean of the finite column abscissae (any value would do: it only conditions the moments); one workgroup
__global__ __launch_bounds__(1024) void k_nesz_center(const double *__restrict__ col, double *__restrict__ x0, long long samples)
{
    __shared__ double sh[2][16];
    double s = 0.0, c = 0.0;
    for (long long k = threadIdx.x; k < samples; k += blockDim.x) {
        const double x = col[samples + k];
        if (isfinite(x)) { s += x; c += 1.0; }
    }
    s = wave_sum_d(s); c = wave_sum_d(c);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ts = 0.0, tc = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { ts += sh[0][w]; tc += sh[1][w]; }
        *x0 = tc > 0.0 ? ts / tc : 0.0;
    }
}

// log10 / exp10 for the arguments of this kernel (positive normal finite doubles; |t| < 300), ~1e-15 relative: frexp + atanh
// series, and 2^k * exp(g) with a degree-13 Taylor polynomial.  About 55 float64 instructions for the pair instead of the
// ~120 of the library calls, which bound the float64 passes (one of each per pixel); anything else goes to the library.
__device__ __forceinline__ double nesz_log10(double v)
{
    if (!(v >= 2.2250738585072014e-308 && v <= 1.7976931348623157e308)) return log10(v);  // 0, negative, denormal, inf, NaN
    int e;
    double m = frexp(v, &e);  // [0.5, 1)
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }  // [sqrt(1/2), sqrt(2))
    // z = (m - 1) / (m + 1) by reciprocal + one Newton step (the IEEE division sequence costs more than the series)
    const double den = m + 1.0;
    double rc = __builtin_amdgcn_rcp(den);
    rc = fma(fma(-den, rc, 1.0), rc, rc);
    rc = fma(fma(-den, rc, 1.0), rc, rc);
    const double z = (m - 1.0) * rc, z2 = z * z;  // |z| <= 0.1716
    double p = 1.0 / 21.0;
    p = fma(p, z2, 1.0 / 19.0); p = fma(p, z2, 1.0 / 17.0); p = fma(p, z2, 1.0 / 15.0); p = fma(p, z2, 1.0 / 13.0);
    p = fma(p, z2, 1.0 / 11.0); p = fma(p, z2, 1.0 / 9.0); p = fma(p, z2, 1.0 / 7.0); p = fma(p, z2, 1.0 / 5.0);
    p = fma(p, z2, 1.0 / 3.0);
    const double lnm = fma(2.0 * z * z2, p, 2.0 * z);  // ln(m) = 2 atanh(z)
    // log10(v) = e log10(2) + ln(m) / ln(10), log10(2) split so that e * hi is exact
    return fma((double)e, 0.30102999566361177, fma((double)e, 3.694239077158931e-13, lnm * 0.4342944819032518));  // hi has 40 bits: e * hi is exact
}
__device__ __forceinline__ double nesz_exp10(double t)
{
    if (!(fabs(t) < 300.0)) return exp10(t);
    const double kf = rint(t * 3.321928094887362);  // t log2(10)
    // g = (t - k log10(2)) ln(10), log10(2) in two parts (k * hi exact for |k| < 2^11)
    const double r = fma(-kf, 3.694239077158931e-13, fma(-kf, 0.30102999566361177, t));
    const double g = r * 2.302585092994046;  // |g| <= 0.3466
    double p = 1.0 / 6227020800.0;
    p = fma(p, g, 1.0 / 479001600.0); p = fma(p, g, 1.0 / 39916800.0); p = fma(p, g, 1.0 / 3628800.0); p = fma(p, g, 1.0 / 362880.0);
    p = fma(p, g, 1.0 / 40320.0); p = fma(p, g, 1.0 / 5040.0); p = fma(p, g, 1.0 / 720.0); p = fma(p, g, 1.0 / 120.0);
    p = fma(p, g, 1.0 / 24.0); p = fma(p, g, 1.0 / 6.0); p = fma(p, g, 0.5); p = fma(p, g, 1.0); p = fma(p, g, 1.0);
    return ldexp(p, (int)kf);
}

// XSW_NESZ_LINES lines per workgroup: the column means and abscissae (float64, L2-resident) are read once per sample and
// serve all of them -- with one line per workgroup they were 24 of the 36 bytes a pixel moved through L1.
#ifndef XSW_NESZ_LINES
#define XSW_NESZ_LINES 4
#endif
#ifndef XSW_NESZ_F32_GROUP
#define XSW_NESZ_F32_GROUP 2  // float32 samples per lane and trip
#endif
#ifndef XSW_NESZ_THREADS
#define XSW_NESZ_THREADS 256
#endif
// The fit of XSW_NESZ_LINES lines per workgroup: (slope, intercept) -> fit[line][2]; k_nesz_eval then writes the raster.  A
// pure read pass followed by a pure write pass runs faster than one kernel doing both per line (measured at 20000^2 float32,
// round 3: fit 0.38 ms + eval 0.60 ms against 1.16 ms).
template <typename T>
__global__ __launch_bounds__(XSW_NESZ_THREADS) void k_nesz_fit(const T *__restrict__ noise, const double *__restrict__ col,
                                                  const double *__restrict__ x0p, double *__restrict__ fit,
                                                  long long lines, long long samples)
{
    constexpr int R = XSW_NESZ_LINES;
    constexpr int NW = XSW_NESZ_THREADS / 64;
    __shared__ double sh[5][NW][R];
    const long long l0 = (long long)blockIdx.x * R;
    const double *mean = col, *xs = col + samples;
    const double x0 = *x0p;
    const T *row[R];
    bool live[R];
#pragma unroll
    for (int j = 0; j < R; ++j) { live[j] = l0 + j < lines; row[j] = noise + (live[j] ? l0 + j : l0) * samples; }
    double n[R], sx[R], sy[R], sxx[R], sxy[R];
#pragma unroll
    for (int j = 0; j < R; ++j) n[j] = sx[j] = sy[j] = sxx[j] = sxy[j] = 0.0;
    // float32 rasters: the reference takes log10 and 10** in float32 (numpy keeps the raster's dtype, utils.py:133-152), so the
    // hardware's float32 log2 / exp2 (v_log_f32 / v_exp_f32, ~1 ulp) are the matching precision -- one instruction each instead
    // of ~50 float64 instructions for the series pair, which is what bounded this kernel (0.38 of the HBM rate in round 2).
    // The moments and the fit stay float64.  float64 rasters keep the float64 series (1e-13 vs numpy).
    constexpr bool F32 = sizeof(T) == 4;
    auto take = [&](int j, double v, double m, double x) {
        if (v != v) v = m;
        // NaN for v < 0 or NaN, -inf for 0: dropped like the reference's isfinite mask
        const double y = F32 ? (double)(3.0102999566398120f * __builtin_amdgcn_logf((float)v)) : 10.0 * nesz_log10(v);
        // polyfit sees x[ok]: a NaN abscissa (column without valid incidence) poisons the fit there; here too
        if (isfinite(y)) { n[j] += 1.0; sx[j] += x; sy[j] += y; sxx[j] += x * x; sxy[j] += x * y; }
    };
    // V samples per lane and trip (adjacent: one 16-byte access when the lines are aligned), R lines each, then the tail
    constexpr int V = (sizeof(T) == 4 ? XSW_NESZ_F32_GROUP : 2);
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const long long groups = samples / V;
    const bool al = (samples % V) == 0 && (((size_t)noise) & (sizeof(vec_t) - 1)) == 0;
    for (long long q = threadIdx.x; q < groups; q += blockDim.x) {
        const long long s = q * V;
        double m[V], xc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { m[k] = mean[s + k]; xc[k] = xs[s + k] - x0; }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            T v[V];
            if (al) { const vec_t t = *(const vec_t *)(row[j] + s);
#pragma unroll
                for (int k = 0; k < V; ++k) v[k] = t[k]; }
            else {
#pragma unroll
                for (int k = 0; k < V; ++k) v[k] = row[j][s + k]; }
#pragma unroll
            for (int k = 0; k < V; ++k) take(j, (double)v[k], m[k], xc[k]);
        }
    }
    for (long long s = groups * V + threadIdx.x; s < samples; s += blockDim.x)
#pragma unroll
        for (int j = 0; j < R; ++j) take(j, (double)row[j][s], mean[s], xs[s] - x0);
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        n[j] = wave_sum_d(n[j]); sx[j] = wave_sum_d(sx[j]); sy[j] = wave_sum_d(sy[j]); sxx[j] = wave_sum_d(sxx[j]); sxy[j] = wave_sum_d(sxy[j]);
        if ((threadIdx.x & 63) == 0) { sh[0][w][j] = n[j]; sh[1][w][j] = sx[j]; sh[2][w][j] = sy[j]; sh[3][w][j] = sxx[j]; sh[4][w][j] = sxy[j]; }
    }
    __syncthreads();
    const double nan = __builtin_nan("");
    double slope[R], icpt[R];  // in the uncentred abscissa: y = slope * x_raw + icpt
#pragma unroll
    for (int j = 0; j < R; ++j) {
        double nn = 0.0, ssx = 0.0, ssy = 0.0, ssxx = 0.0, ssxy = 0.0;
#pragma unroll
        for (int k = 0; k < NW; ++k) { nn += sh[0][k][j]; ssx += sh[1][k][j]; ssy += sh[2][k][j]; ssxx += sh[3][k][j]; ssxy += sh[4][k][j]; }
        if (nn == 0.0) {  // nothing to fit: the reference returns a NaN line (utils.py:146-149)
            slope[j] = nan; icpt[j] = nan;
        } else {
            const double det = nn * ssxx - ssx * ssx;  // n^2 var(x) >= 0
            const double xm = ssx / nn, ym = ssy / nn;
            if (det > 1e-24 * (nn * ssxx + ssx * ssx + 1e-300)) {
                slope[j] = (nn * ssxy - ssx * ssy) / det;
                icpt[j] = ym - slope[j] * (xm + x0);
            } else {
                // every fitted abscissa equal (one valid sample, or constant incidence): numpy's lstsq returns the
                // minimum-norm solution of the column-scaled rank-1 system, slope = y/(2x), intercept = y/2
                slope[j] = ym / (2.0 * (xm + x0));
                icpt[j] = 0.5 * ym;
            }
        }
    }
    if (threadIdx.x < R && live[threadIdx.x]) {
        double sl = slope[0], ic = icpt[0];
#pragma unroll
        for (int j = 1; j < R; ++j) { sl = (int)threadIdx.x == j ? slope[j] : sl; ic = (int)threadIdx.x == j ? icpt[j] : ic; }
        fit[2 * (l0 + threadIdx.x)] = sl;
        fit[2 * (l0 + threadIdx.x) + 1] = ic;
    }
}

// out[l][s] = 10 ** ((inc_row[s] * slope[l] + icpt[l] - 1) / 10): a thread keeps its EV column abscissae in registers and
// streams down its block of lines (the line's two coefficients are wave-uniform: scalar loads); only the stores touch HBM.
#ifndef XSW_NESZ_EV
#define XSW_NESZ_EV 2
#endif
template <bool F32>
__global__ __launch_bounds__(256) void k_nesz_eval(const double *__restrict__ col, const double *__restrict__ fit, double *__restrict__ out,
                                                   long long lines, long long samples, long long lines_per_block)
{
    constexpr int EV = XSW_NESZ_EV;
    const long long s = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * EV;
    if (s >= samples) return;
    const long long l0 = (long long)blockIdx.y * lines_per_block;
    const long long l1 = l0 + lines_per_block < lines ? l0 + lines_per_block : lines;
    const double *xs = col + samples;
    double x[EV];
#pragma unroll
    for (int k = 0; k < EV; ++k) x[k] = xs[s + k < samples ? s + k : samples - 1];
    const bool vec = s + EV <= samples && (samples & 1) == 0 && (((size_t)out) & 15) == 0;
    double *o = out + l0 * samples + s;
    for (long long l = l0; l < l1; ++l, o += samples) {
        const double sl = fit[2 * l], ic = fit[2 * l + 1];
        double r[EV];
#pragma unroll
        for (int k = 0; k < EV; ++k) {
            const double t = (x[k] * sl + ic - 1.0) * 0.1;
            r[k] = F32 ? (double)__builtin_amdgcn_exp2f((float)(t * 3.321928094887362)) : nesz_exp10(t);
        }
        if (vec) {
#pragma unroll
            for (int k = 0; k < EV; k += 2) { double2 t; t.x = r[k]; t.y = r[k + 1]; *(double2 *)(o + k) = t; }
        } else {
#pragma unroll
            for (int k = 0; k < EV; ++k)
                if (s + k < samples) o[k] = r[k];
        }
    }
}

}  // namespace xsw
