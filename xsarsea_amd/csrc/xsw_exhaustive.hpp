// Exhaustive (wspd x phi) sweep with the LUT slice tiled through LDS -- the literal form of the
// reference's search (windspeed.py:220-229: every candidate of the incidence slice is scored) laid
// out for a CDNA4 compute unit:
//
//   * a workgroup (4 waves) owns a 2-D raster tile of 4 lines x 64 samples: incidence varies almost
//     only along `sample`, so the 256 pixels of a tile fall in one or two 0.1-degree bins and share
//     the LUT slice;
//   * for each distinct bin present in the tile, the slice is streamed through LDS in chunks of
//     `rows_per_chunk` wind speeds (coalesced 16-B global loads -> ds_write_b128), once per workgroup;
//   * each wave walks its pixels of that bin; the pixel's parameters are wave-uniform (SGPRs), the 64
//     lanes sweep the chunk's candidates out of LDS (conflict-free ds_read_b64, lane = direction) and
//     keep (best, second-best, index); a wave-level butterfly reduction merges them and the owning
//     lane folds the chunk result into the pixel's running state;
//   * a pixel whose second-best screening score is within eps of its best is re-done by the exact
//     full scan (reference operation order); otherwise the best is the reference's argmin.
//
// Mono co-pol only (the benchmark configuration); uniform finite LUTs only (host checks).
#pragma once
#include "xsw_device.hpp"

namespace xsw {

struct BestSecond {
    double b, s;
    int i;
};

__device__ __forceinline__ void merge_bs(BestSecond &x, double ob, double os, int oi)
{
    const double mx = fmax(x.b, ob);
    const double ns = fmin(fmin(x.s, os), mx);
    x.i = (ob < x.b) ? oi : x.i;
    x.b = fmin(x.b, ob);
    x.s = ns;
}

__device__ __forceinline__ void wave_merge_bs(BestSecond &x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(x.b, off), os = __shfl_xor(x.s, off);
        const int oi = __shfl_xor(x.i, off);
        merge_bs(x, ob, os, oi);
    }
}

template <typename T, typename TO>
__global__ __launch_bounds__(256) void k_invert_exhaustive(DevTables L, KArgs A, int rows_per_chunk)
{
    extern __shared__ __align__(16) double lds_chunk[];  // [rows_per_chunk][phi_pad]
    __shared__ int sh_bin;

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // tile = 4 lines x 64 samples
    const long long strips_per_line = (A.samples + 63) / 64;
    const long long tile_line = (long long)(blockIdx.x / strips_per_line) * 4 + wv;
    const long long tile_s0 = (long long)(blockIdx.x % strips_per_line) * 64;
    const long long smp = tile_s0 + lane;
    const bool in = tile_line < A.lines && smp < A.samples;
    const long long i = in ? tile_line * A.samples + smp : 0;

    Pixel P;
    load_pixel<T>(L, A, i, in, P);
    bool pending = (P.flags & F_NEED_CO) != 0;
    BestSecond run;
    run.b = __builtin_inf(); run.s = __builtin_inf(); run.i = 0x7fffffff;
    const double ah = 0.5 * P.a_re, bh = 0.5 * P.b_eff;
    const double sn = -P.s_co * A.inv_dsig_co;
    unsigned cand = 0;

    for (;;) {
        // next distinct incidence bin of the tile (block-uniform)
        if (threadIdx.x == 0) sh_bin = 0x7fffffff;
        __syncthreads();
        int mb = wave_min_i(pending ? P.i_inc : 0x7fffffff);
        if (lane == 0 && mb != 0x7fffffff) atomicMin(&sh_bin, mb);
        __syncthreads();
        const int cur = sh_bin;
        __syncthreads();
        if (cur == 0x7fffffff) break;
        const double *__restrict__ slice = L.co + (size_t)cur * L.n_w * L.phi_pad;
        const bool mine = pending && P.i_inc == cur;
        const unsigned long long todo0 = __ballot(mine);

        for (int r0 = 0; r0 < L.n_w; r0 += rows_per_chunk) {
            const int rows = min(rows_per_chunk, L.n_w - r0);
            // stage chunk: contiguous rows*phi_pad doubles, 16-B vectors (phi_pad % 4 == 0)
            const double2 *__restrict__ src = (const double2 *)(slice + (size_t)r0 * L.phi_pad);
            double2 *dst = (double2 *)lds_chunk;
            const int nvec = rows * L.phi_pad / 2;
            for (int v = threadIdx.x; v < nvec; v += 256) dst[v] = src[v];
            __syncthreads();

            unsigned long long todo = todo0;
            while (todo) {
                const int p = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const double uah = rd_lane_d(ah, p), ubh = rd_lane_d(bh, p), usn = rd_lane_d(sn, p);
                BestSecond x;
                x.b = __builtin_inf(); x.s = __builtin_inf(); x.i = 0x7fffffff;
                for (int c0 = 0; c0 < L.n_phi; c0 += 64) {
                    const int ip = c0 + lane;
                    const bool ok = ip < L.n_phi;
                    const int ipc = ok ? ip : 0;
                    const double U = 2.0 * (uah * L.cphi[ipc] + ubh * L.sphi[ipc]);
                    const double *col = lds_chunk + ipc;
                    for (int r = 0; r < rows; ++r) {
                        const double wh = L.wh[r0 + r];
                        const double dd = fma(col[r * L.phi_pad], A.inv_dsig_co, usn);
                        double J = fma(dd, dd, wh * (wh - U));
                        J = ok ? J : __builtin_inf();
                        x.s = fmin(x.s, fmax(J, x.b));
                        if (J < x.b) { x.b = J; x.i = (r0 + r) * L.n_phi + ipc; }
                    }
                }
                wave_merge_bs(x);
                if (lane == p) merge_bs(run, x.b, x.s, x.i);
                cand += (unsigned)(rows * L.n_phi);
            }
            __syncthreads();  // chunk consumed before it is overwritten
        }
        if (mine) pending = false;
    }

    // settle: unique screening minimum == reference argmin; otherwise exact full scan of that pixel
    const double m2 = ah * ah + bh * bh;
    const double Tthr = run.b + 1e-9 * (1.0 + fabs(run.b) + m2);
    const bool need = (P.flags & F_NEED_CO) != 0;
    const bool ambiguous = need && (!(P.flags & F_CO_FINITE) || !(run.b < __builtin_inf()) || run.s <= Tthr);
    int my_flat = run.i;
    unsigned n_exact = 0;
    unsigned long long amb = __ballot(ambiguous);
    while (amb) {
        const int p = __ffsll((long long)amb) - 1;
        amb &= amb - 1;
        const int flat = exact_scan_co(L, rd_lane_i(P.i_inc, p), rd_lane_d(P.s_co, p), rd_lane_d(P.a_re, p),
                                       rd_lane_d(P.b_eff, p), A.dsig_co, lane);
        if (lane == p) my_flat = flat;
        n_exact++;
    }
    if (A.stats && lane == 0) {
        atomicAdd(&A.stats[0], (unsigned long long)__popcll(__ballot(need)));
        atomicAdd(&A.stats[1], (unsigned long long)cand);
        atomicAdd(&A.stats[2], (unsigned long long)n_exact);
    }
    if (in) store_pixel<TO>(L, A, i, P, my_flat, -1);
}

template <typename T, typename TO>
static hipError_t launch_exhaustive(const DevTables &L, const KArgs &A, hipStream_t stream)
{
    const size_t lds_budget = 64 * 1024;  // two workgroups per CU
    int rows = (int)(lds_budget / ((size_t)L.phi_pad * sizeof(double)));
    if (rows < 1) return hipErrorInvalidValue;
    if (rows > L.n_w) rows = L.n_w;
    const size_t lds = (size_t)rows * L.phi_pad * sizeof(double);
    const long long strips_per_line = (A.samples + 63) / 64;
    const long long nblocks = strips_per_line * ((A.lines + 3) / 4);
    if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_invert_exhaustive<T, TO>), dim3((unsigned)nblocks), dim3(256), lds, stream, L, A, rows);
    return hipGetLastError();
}

}  // namespace xsw
