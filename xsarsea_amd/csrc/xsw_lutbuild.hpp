// Device side of the LUT preparation (reference: GmfModel._raw_lut, gmfs.py:350-395; Model._normalize_lut,
// models.py:142-168; the dB conversion of Model.to_lut, models.py:210-216; the closure arrays of
// _invert_from_model_numpy, windspeed.py:144-181).  With these kernels a built-in GMF goes from its name to a searchable
// table without the 362 MB float64 block ever visiting the host:
//   k_gmf_grid   raw[i][j][k] = gmf(inc[i], wspd[j], phi[k])          (the fill of the raw grid, linear units)
//   k_lut_interp low -> high resolution (xsw_device.hpp)
//   k_to_db      10 log10(x + 1e-15) in place
//   k_pad_co     dense [n_inc][n_wspd][n_phi] -> phi-padded float64 + float32 copies, finiteness flag, max |dB|
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xsw_gmf.hpp"
#include "xsw_device.hpp"

namespace xsw {

template <int M>
__global__ __launch_bounds__(256) void k_gmf_grid(int id, const double *__restrict__ inc, const double *__restrict__ wspd,
                                                  const double *__restrict__ phi, int ni, int nw, int np, double *__restrict__ out)
{
    const int npp = np > 0 ? np : 1;
    const long long n = (long long)ni * nw * npp;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(t % npp), j = (int)((t / npp) % nw), i = (int)(t / ((long long)npp * nw));
        out[t] = gmf_eval<M>(id, inc[i], wspd[j], np > 0 ? phi[k] : 0.0);
    }
}

__global__ __launch_bounds__(256) void k_to_db(double *__restrict__ x, long long n)
{
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x)
        x[t] = 10.0 * log10(x[t] + 1e-15);
}

// One wave per LUT row (n_phi values -> phi_pad slots, the pad zero-filled); flags[0] |= 1 when a value is not finite,
// flags[1] = bits of max |finite value| (non-negative doubles order like their bit patterns).
__global__ __launch_bounds__(256) void k_pad_co(const double *__restrict__ dense, double *__restrict__ co,
                                                float *__restrict__ co32, int n_phi, int phi_pad, long long rows,
                                                unsigned long long *__restrict__ flags)
{
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    double amax = 0.0;
    bool bad = false;
    for (long long r = wave0; r < rows; r += nwaves) {
        const double *src = dense + r * n_phi;
        double *dst = co + r * phi_pad;
        float *dst32 = co32 + r * phi_pad;
        for (int k = lane; k < phi_pad; k += 64) {
            const double v = k < n_phi ? src[k] : 0.0;
            dst[k] = v;
            dst32[k] = (float)v;
            const double a = fabs(v);
            if (a <= 1.79769313486231570815e308) amax = fmax(amax, a); else bad = true;  // NaN and inf fail the test
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off));
    const unsigned long long any_bad = __ballot(bad);
    if (lane == 0) {
        if (any_bad) atomicOr(&flags[0], 1ULL);
        atomicMax(&flags[1], (unsigned long long)__double_as_longlong(amax));
    }
}

// out[i] = min over the directions of slice i of (index of the first row that is LOWER than its predecessor), n_w when
// none: every column of slice i is non-decreasing in wind speed over rows [0, out[i]) -- the precondition of the band
// pruning of co_band_pass (CMOD5.N itself saturates and decreases beyond 24..40 m/s at incidences below 41 deg).
// out[] preset to n_w by the host.
__global__ __launch_bounds__(256) void k_mono_rows(const double *__restrict__ dense, int n_inc, int n_w, int n_phi, int *__restrict__ out)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n_inc * n_phi) return;
    const int i = (int)(t / n_phi), p = (int)(t % n_phi);
    const double *col = dense + (size_t)i * n_w * n_phi + p;
    int first = n_w;
    double prev = col[0];
    for (int r = 1; r < n_w; ++r) {
        const double v = col[(size_t)r * n_phi];
        if (v < prev) { first = r; break; }
        prev = v;
    }
    if (first < n_w) atomicMin(&out[i], first);
}

// Tail minima (band_wave's tail cut): tail[i][k][p] = the smallest value of slice i in rows >= mono[i] over the directions
// p .. p + 2^k - 1 (clipped to the axis), k = 0 .. XSW_TAIL_LEVELS - 1 -- a sparse table, so that the minimum over a window's
// directions [a, b] is min(tail[k][a], tail[k][b - 2^k + 1]) with 2^k <= b - a + 1 < 2^(k+1) -- and level XSW_TAIL_LEVELS holds
// the minimum over ALL directions (windows wider than 2^XSW_TAIL_LEVELS - 1).  +inf where every row is monotone.  A pixel whose
// band ends below the minimum of its window's directions has no candidate in the rows past the monotone ones.
// One workgroup per slice.
__global__ __launch_bounds__(256) void k_tail_min(const double *__restrict__ dense, int n_w, int n_phi, int phi_pad, const int *__restrict__ mono,
                                                  double *__restrict__ tail)
{
    const int i = blockIdx.x, m = mono[i];
    const double *sl = dense + (size_t)i * n_w * n_phi;
    double *out = tail + (size_t)i * (XSW_TAIL_LEVELS + 1) * phi_pad;
    const double inf = __builtin_inf();
    for (int p = threadIdx.x; p < phi_pad; p += blockDim.x) {
        double lo = inf;
        if (p < n_phi)
            for (int r = m; r < n_w; ++r) lo = fmin(lo, sl[(size_t)r * n_phi + p]);
        out[p] = lo;
    }
    __syncthreads();
    for (int k = 1; k < XSW_TAIL_LEVELS; ++k) {
        const double *prev = out + (size_t)(k - 1) * phi_pad;
        double *cur = out + (size_t)k * phi_pad;
        const int h = 1 << (k - 1);
        for (int p = threadIdx.x; p < phi_pad; p += blockDim.x) cur[p] = fmin(prev[p], p + h < phi_pad ? prev[p + h] : inf);
        __syncthreads();
    }
    __shared__ double slo[256];
    double lo = inf;
    for (int p = threadIdx.x; p < n_phi; p += blockDim.x) lo = fmin(lo, out[p]);
    slo[threadIdx.x] = lo;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) slo[threadIdx.x] = fmin(slo[threadIdx.x], slo[threadIdx.x + st]);
        __syncthreads();
    }
    const double all = slo[0];
    for (int p = threadIdx.x; p < phi_pad; p += blockDim.x) out[(size_t)XSW_TAIL_LEVELS * phi_pad + p] = all;
}

// dB range of the monotone rows of each slice -> the slice's uniform threshold grid {t0, width, 1 / width}
__global__ __launch_bounds__(256) void k_inv_range(const double *__restrict__ dense, int n_w, int n_phi, const int *__restrict__ mono,
                                                   double *__restrict__ grid)
{
    const int i = blockIdx.x;
    const double *sl = dense + (size_t)i * n_w * n_phi;
    const long long n = (long long)mono[i] * n_phi;
    double lo = __builtin_inf(), hi = -__builtin_inf();
    for (long long k = threadIdx.x; k < n; k += blockDim.x) { const double v = sl[k]; lo = fmin(lo, v); hi = fmax(hi, v); }
    __shared__ double slo[256], shi[256];
    slo[threadIdx.x] = lo; shi[threadIdx.x] = hi;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) { slo[threadIdx.x] = fmin(slo[threadIdx.x], slo[threadIdx.x + st]); shi[threadIdx.x] = fmax(shi[threadIdx.x], shi[threadIdx.x + st]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double width = (shi[0] - slo[0]) / (double)XSW_INV_BINS;
        const bool ok = width > 0.0 && width < 1e300 && slo[0] > -1e300;
        grid[3 * i + 0] = ok ? slo[0] : 0.0;
        grid[3 * i + 1] = ok ? width : 0.0;
        grid[3 * i + 2] = ok ? 1.0 / width : 0.0;  // not ok (flat or non-finite slice): every threshold falls in bin 0, whose rows are 0
    }
}
// one thread per (slice, direction): merge of the ascending thresholds with the non-decreasing column
__global__ __launch_bounds__(256) void k_inv_rows(const double *__restrict__ dense, int n_inc, int n_w, int n_phi, int phi_pad,
                                                  const int *__restrict__ mono, const double *__restrict__ grid,
                                                  unsigned short *__restrict__ inv)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n_inc * n_phi) return;
    const int i = (int)(t / n_phi), p = (int)(t % n_phi);
    const double *col = dense + (size_t)i * n_w * n_phi + p;
    unsigned short *out = inv + (size_t)i * XSW_INV_BINS * phi_pad + p;
    const double t0 = grid[3 * i], width = grid[3 * i + 1];
    const int m = mono[i];
    int r = 0;
    out[0] = 0;  // bin 0 also serves thresholds below the grid: row 0 is a lower estimate of any lower bound
    for (int b = 1; b < XSW_INV_BINS; ++b) {
        const double thr = fma((double)b, width, t0);
        while (r < m && col[(size_t)r * n_phi] < thr) ++r;
        out[(size_t)b * phi_pad] = (unsigned short)r;
    }
}

// ---- block pyramid (co_block_search, xsw_device.hpp): min / max of the LUT per block of XSW_BLK_R speed rows x XSW_BLK_C
// directions, float32 rounded OUTWARD (a bound must never be tighter than the table), and per band of `g` block rows over all
// directions.  One thread per block / per band.
__global__ __launch_bounds__(256) void k_block_minmax(const double *__restrict__ dense, int n_inc, int n_w, int n_phi, int nbr, int nbc,
                                                      float2 *__restrict__ blk, int blk_c = XSW_BLK_C /* directions per block: XSW_BLK_C, XSW_BLK_C4 for the sub-blocks, XSW_CELL_C XSW_BLK_C for the cells */,
                                                      int blk_r = XSW_BLK_R /* rows per block (XSW_CELL_R XSW_BLK_R for the cells) */)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n_inc * nbr * nbc) return;
    const int bc = (int)(t % nbc), br = (int)((t / nbc) % nbr), i = (int)(t / ((long long)nbc * nbr));
    const double *sl = dense + (size_t)i * n_w * n_phi;
    double lo = __builtin_inf(), hi = -__builtin_inf();
    for (int r = br * blk_r; r < min(br * blk_r + blk_r, n_w); ++r)
        for (int c = bc * blk_c; c < min(bc * blk_c + blk_c, n_phi); ++c) {
            const double v = sl[(size_t)r * n_phi + c];
            lo = fmin(lo, v);
            hi = fmax(hi, v);
        }
    float2 o;
    o.x = __double2float_rd(lo);
    o.y = __double2float_ru(hi);
    blk[t] = o;
}
__global__ __launch_bounds__(256) void k_band_minmax(const float2 *__restrict__ blk, int n_inc, int nbr, int nbc, int g, int nbands,
                                                     float2 *__restrict__ band)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n_inc * nbands) return;
    const int tb = (int)(t % nbands), i = (int)(t / nbands);
    const float2 *b = blk + (size_t)i * nbr * nbc;
    float lo = __builtin_inff(), hi = -__builtin_inff();
    for (int br = tb * g; br < min((tb + 1) * g, nbr); ++br)
        for (int bc = 0; bc < nbc; ++bc) {
            const float2 v = b[br * nbc + bc];
            lo = fminf(lo, v.x);
            hi = fmaxf(hi, v.y);
        }
    float2 o;
    o.x = lo;
    o.y = hi;
    band[t] = o;
}

}  // namespace xsw
