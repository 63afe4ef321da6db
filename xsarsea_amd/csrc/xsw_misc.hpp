// Kernels of libxsw that are not part of the search: LUT transposition and resolution change, sigma0_detrend's divide.
// Included by xsw.hip only (the search kernels live in xsw_device.hpp / xsw_band.hpp / xsw_exhaustive.hpp).
#pragma once
#include "xsw_device.hpp"

namespace xsw {

// [n_inc][n_w][phi_pad] -> [n_inc][n_phi][w_pad], 32x32 LDS tiles
__global__ __launch_bounds__(256) void k_transpose_slices(const double *__restrict__ src, double *__restrict__ dst,
                                                           int n_w, int n_phi, int phi_pad, int w_pad)
{
    __shared__ double tile[32][33];
    const int s = blockIdx.z;
    const double *in = src + (size_t)s * n_w * phi_pad;
    double *out = dst + (size_t)s * n_phi * w_pad;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int p0 = blockIdx.x * 32, w0 = blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8) {
        int w = w0 + j, p = p0 + tx;
        tile[j][tx] = (w < n_w && p < n_phi) ? in[(size_t)w * phi_pad + p] : 0.0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int p = p0 + j, w = w0 + tx;
        if (p < n_phi && w < n_w) out[(size_t)p * w_pad + w] = tile[tx][j];
    }
}

// LUT resolution change (models.py:142-168): out[i][j][k] = lerp_phi(lerp_wspd(lerp_inc(raw))) with the
// staged rounding of three sequential interp1d passes.  lo*[] hold, per target point, the index of the left
// raw neighbour (searchsorted(...).clip(1, n-1) - 1, computed on the host).  n_phi == 0: 2-D table.
struct InterpArgs {
    const double *raw, *xi_raw, *xw_raw, *xp_raw, *xi, *xw, *xp;
    const int *loi, *low, *lop;
    double *out;
    int ni_raw, nw_raw, np_raw, ni, nw, np;
};
__device__ __forceinline__ double lerp1(double y_lo, double y_hi, double x_lo, double x_hi, double x)
{
    const double slope = (y_hi - y_lo) / (x_hi - x_lo);
    return slope * (x - x_lo) + y_lo;
}
__global__ __launch_bounds__(256) void k_lut_interp(InterpArgs a)
{
    const int np = a.np > 0 ? a.np : 1, np_raw = a.np_raw > 0 ? a.np_raw : 1;
    const long long n = (long long)a.ni * a.nw * np;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(t % np), j = (int)((t / np) % a.nw), i = (int)(t / ((long long)np * a.nw));
        const int i0 = a.loi[i], j0 = a.low[j];
        const double xi0 = a.xi_raw[i0], xi1 = a.xi_raw[i0 + 1], xi = a.xi[i];
        const double xw0 = a.xw_raw[j0], xw1 = a.xw_raw[j0 + 1], xw = a.xw[j];
        const int nk = a.np > 0 ? 2 : 1;
        const int k0 = a.np > 0 ? a.lop[k] : 0;
        double b[2];
        for (int kk = 0; kk < nk; ++kk) {
            double aa[2];
            for (int jj = 0; jj < 2; ++jj) {
                const size_t o = ((size_t)i0 * a.nw_raw + (j0 + jj)) * np_raw + (k0 + kk);
                aa[jj] = lerp1(a.raw[o], a.raw[o + (size_t)a.nw_raw * np_raw], xi0, xi1, xi);  // incidence pass
            }
            b[kk] = lerp1(aa[0], aa[1], xw0, xw1, xw);                                          // wspd pass
        }
        a.out[t] = a.np > 0 ? lerp1(b[0], b[1], a.xp_raw[k0], a.xp_raw[k0 + 1], a.xp[k]) : b[0];  // phi pass
    }
}

// sigma0_detrend's per-pixel work (detrend.py:64): out = sigma0 / ratio[sample].  Purely HBM-bound:
// 16-B vector loads/stores (4 samples per thread), the ratio row stays in L2; no integer division per pixel.
// streaming accesses of k_detrend: every byte is touched once.  Non-temporal hints measured no gain on MI355X
// (bit 0 = loads, bit 1 = stores: f32->f64 1.00 ms plain / 1.00 ms nt loads / 1.40 ms nt loads+stores), so plain.
#ifndef XSW_DETREND_LINES
#define XSW_DETREND_LINES 4
#endif
#define XSW_DETREND_LD(p) (*(p))
#define XSW_DETREND_ST(v, p) (*(p) = (v))
template <typename T, int N> struct VecOf;
template <> struct VecOf<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct VecOf<double, 4> { typedef double type __attribute__((ext_vector_type(4))); };

// FAST = 1: x / r as q0 = x*y, q = fma(fma(-q0, r, x), y, q0) with y = RN(1/r) prepared on the host: the
// correctly rounded quotient (Markstein) whenever r is finite, non-zero, within 2^+-500 and its significand is not
// all ones -- the host checks every r and otherwise launches FAST = 0 (IEEE division sequence, ~10x the VALU work).
// Non-finite q0 (x = +-inf or NaN) is returned as is, which is what the division gives.
template <int FAST>
__device__ __forceinline__ double div_by(double x, double r, double y)
{
    if (!FAST) return x / r;
    const double q0 = x * y;
    const double q1 = fma(fma(-q0, r, x), y, q0);
    return isfinite(q0) ? q1 : q0;
}

// grid.x tiles the sample axis in quads (4 samples per thread, 16-B accesses), grid.y tiles the lines; a thread
// keeps its 4 divisors (and reciprocals) in registers and streams down its lines: per pixel only the sigma0 load
// and the store touch memory.  Samples not divisible by 4: the last (partial) quad is handled element-wise.
template <typename T, typename TO, int FAST>
__global__ __launch_bounds__(256) void k_detrend(const T *__restrict__ sigma0, const double *__restrict__ ratio,
                                                 const double *__restrict__ rinv, TO *__restrict__ out, long long lines,
                                                 long long samples, long long lines_per_block)
{
    const long long quad = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long s0 = quad * 4;
    if (s0 >= samples) return;
    const long long l0 = (long long)blockIdx.y * lines_per_block;
    const long long l1 = l0 + lines_per_block < lines ? l0 + lines_per_block : lines;
    const bool full = s0 + 4 <= samples && (samples & 3) == 0;  // aligned 16-B accesses need samples % 4 == 0
    double r[4], y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long sk = s0 + k < samples ? s0 + k : samples - 1;
        r[k] = ratio[sk];
        y[k] = FAST ? rinv[sk] : 0.0;
    }
    if (full) {
        typedef typename VecOf<T, 4>::type vin_t;
        typedef typename VecOf<TO, 4>::type vout_t;
        const vin_t *in = (const vin_t *)(sigma0 + l0 * samples + s0);
        vout_t *o = (vout_t *)(out + l0 * samples + s0);
        const long long stride = samples >> 2;
        long long l = l0;
        for (; l + XSW_DETREND_LINES <= l1; l += XSW_DETREND_LINES) {  // several lines in flight per lane
            vin_t a[XSW_DETREND_LINES];
#pragma unroll
            for (int u = 0; u < XSW_DETREND_LINES; ++u) a[u] = XSW_DETREND_LD(&in[u * stride]);
#pragma unroll
            for (int u = 0; u < XSW_DETREND_LINES; ++u) {
                vout_t ou;
                ou.x = (TO)div_by<FAST>((double)a[u].x, r[0], y[0]); ou.y = (TO)div_by<FAST>((double)a[u].y, r[1], y[1]);
                ou.z = (TO)div_by<FAST>((double)a[u].z, r[2], y[2]); ou.w = (TO)div_by<FAST>((double)a[u].w, r[3], y[3]);
                XSW_DETREND_ST(ou, &o[u * stride]);
            }
            in += XSW_DETREND_LINES * stride; o += XSW_DETREND_LINES * stride;
        }
        for (; l < l1; ++l) {
            const vin_t a = in[0];
            vout_t oa;
            oa.x = (TO)div_by<FAST>((double)a.x, r[0], y[0]); oa.y = (TO)div_by<FAST>((double)a.y, r[1], y[1]);
            oa.z = (TO)div_by<FAST>((double)a.z, r[2], y[2]); oa.w = (TO)div_by<FAST>((double)a.w, r[3], y[3]);
            o[0] = oa;
            in += stride; o += stride;
        }
    } else {
        for (long long l = l0; l < l1; ++l)
            for (int k = 0; k < 4 && s0 + k < samples; ++k)
                out[l * samples + s0 + k] = (TO)div_by<FAST>((double)sigma0[l * samples + s0 + k], r[k], y[k]);
    }
}

}  // namespace xsw
