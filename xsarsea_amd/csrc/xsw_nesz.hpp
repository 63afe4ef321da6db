// Cross-pol noise flattening on the device (reference: windspeed/utils.py:94-163, `nesz_flattening`): the full-raster
// pass that precedes the dual-pol inversion in the L1 workflow.
//
//   noise_mean[s] = nanmean(noise[:, s])           inc_row[s] = nanmean(inc[:, s])                     (:119-121, :163)
//   per line l:  filled = where(isnan(noise[l]), noise_mean, noise[l]);  y = 10 log10(filled);  ok = isfinite(y)
//                (slope, icpt) = degree-1 least squares of y[ok] against inc_row[ok]                    (:133-149)
//                out[l][s] = 10 ** ((inc_row[s] * slope + icpt - 1) / 10)   for EVERY s                (:152)
//
// Three kernels, all HBM-bound streaming passes (float32 rasters: 8 B read in the first, 4 B read + 8 B written in the
// third, per pixel):
//   k_nesz_colsum   partial column sums/counts over blocks of lines (one or four columns per thread, coalesced rows)
//   k_nesz_colmean  finishes the column means in line-block order (deterministic) + the centring abscissa x0
//   k_nesz_rows     one workgroup per line: five float64 moments about x0 (n, Sx, Sy, Sxx, Sxy), closed-form fit,
//                   then the line's outputs from the column abscissae (no second read of the raster).
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

template <typename T>
__global__ __launch_bounds__(256) void k_nesz_colsum(const T *__restrict__ noise, const T *__restrict__ inc,
                                                     NeszPartial *__restrict__ part, long long lines, long long samples,
                                                     long long lines_per_block)
{
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= samples) return;
    const long long l0 = (long long)blockIdx.y * lines_per_block;
    const long long l1 = l0 + lines_per_block < lines ? l0 + lines_per_block : lines;
    double sn = 0.0, si = 0.0;
    int cn = 0, ci = 0;
    const T *pn = noise + l0 * samples + s, *pi = inc + l0 * samples + s;
    long long l = l0;
    for (; l + 4 <= l1; l += 4) {  // four lines in flight per lane
        T a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = pn[u * samples]; b[u] = pi[u * samples]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double x = (double)a[u], y = (double)b[u];
            if (x == x) { sn += x; ++cn; }
            if (y == y) { si += y; ++ci; }
        }
        pn += 4 * samples; pi += 4 * samples;
    }
    for (; l < l1; ++l) {
        const double x = (double)*pn, y = (double)*pi;
        if (x == x) { sn += x; ++cn; }
        if (y == y) { si += y; ++ci; }
        pn += samples; pi += samples;
    }
    NeszPartial p;
    p.sum_n = sn; p.sum_i = si; p.cnt_n = cn; p.cnt_i = ci;
    part[(long long)blockIdx.y * samples + s] = p;
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

template <typename T>
__global__ __launch_bounds__(256) void k_nesz_rows(const T *__restrict__ noise, const double *__restrict__ col,
                                                   const double *__restrict__ x0p, double *__restrict__ out,
                                                   long long samples)
{
    __shared__ double sh[5][4];
    const long long l = blockIdx.x;
    const T *row = noise + l * samples;
    const double *mean = col, *xs = col + samples;
    const double x0 = *x0p;
    double n = 0.0, sx = 0.0, sy = 0.0, sxx = 0.0, sxy = 0.0;
    for (long long s = threadIdx.x; s < samples; s += blockDim.x) {
        double v = (double)row[s];
        if (v != v) v = mean[s];
        const double y = 10.0 * log10(v);  // NaN for v < 0 or NaN, -inf for 0: dropped like the reference's isfinite mask
        const double x = xs[s] - x0;
        // polyfit sees x[ok]: a NaN abscissa (column without valid incidence) poisons the fit there; here too
        if (isfinite(y)) { n += 1.0; sx += x; sy += y; sxx += x * x; sxy += x * y; }
    }
    n = wave_sum_d(n); sx = wave_sum_d(sx); sy = wave_sum_d(sy); sxx = wave_sum_d(sxx); sxy = wave_sum_d(sxy);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = n; sh[1][w] = sx; sh[2][w] = sy; sh[3][w] = sxx; sh[4][w] = sxy; }
    __syncthreads();
    n = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    sx = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    sy = sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3];
    sxx = sh[3][0] + sh[3][1] + sh[3][2] + sh[3][3];
    sxy = sh[4][0] + sh[4][1] + sh[4][2] + sh[4][3];
    const double nan = __builtin_nan("");
    double slope, icpt;  // in the uncentred abscissa: y = slope * x_raw + icpt
    if (n == 0.0) {  // nothing to fit: the reference returns a NaN line (utils.py:146-149)
        slope = nan; icpt = nan;
    } else {
        const double det = n * sxx - sx * sx;  // n^2 var(x) >= 0
        const double xm = sx / n, ym = sy / n;
        if (det > 1e-24 * (n * sxx + sx * sx + 1e-300)) {
            slope = (n * sxy - sx * sy) / det;
            icpt = ym - slope * (xm + x0);
        } else {
            // every fitted abscissa equal (one valid sample, or constant incidence): numpy's lstsq returns the
            // minimum-norm solution of the column-scaled rank-1 system, slope = y/(2x), intercept = y/2
            const double xr = xm + x0;
            slope = ym / (2.0 * xr);
            icpt = 0.5 * ym;
        }
    }
    double *o = out + l * samples;
    for (long long s = threadIdx.x; s < samples; s += blockDim.x) {
        const double t = (xs[s] * slope + icpt - 1.0) / 10.0;
        o[s] = exp10(t);
    }
}

}  // namespace xsw
