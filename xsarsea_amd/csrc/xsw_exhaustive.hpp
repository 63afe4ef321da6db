// Exhaustive (wspd x phi) sweep with the LUT slice tiled through LDS -- the literal form of the
// reference's search (windspeed.py:220-229: every candidate of the incidence slice is scored) laid
// out for a CDNA4 compute unit:
//
//   * a workgroup (4 waves) owns a 2-D raster tile of 4 lines x 64 samples (same XCD-aware tile walk as
//     k_invert): incidence varies almost only along `sample`, so the 256 pixels of a tile fall in one or two
//     0.1-degree bins and share the LUT slice;
//   * for each distinct bin of the tile, every wave takes its pixels of that bin in batches of 8; for each
//     batch the slice is streamed through LDS in chunks of `rows_per_chunk` wind speeds (coalesced 16-B
//     global loads -> ds_write_b128, once per workgroup per batch; the slice stays hot in the XCD's L2);
//   * inside a chunk the 64 lanes sweep the candidates of each pixel of the batch out of LDS
//     (conflict-free ds_read_b64, lane = direction, rows unrolled by four, no per-candidate masking):
//     score = fma(dd, dd, p) with dd = fma(LUT, 1/dsig, -s/dsig) and p = wh*(wh - U_phi) advanced along the
//     speed axis by forward differences; each lane keeps (best, second best, code of best) per pixel in
//     registers across all chunks, so the wave-level argmin (DPP butterfly) runs once per pixel;
//   * a pixel whose screening minimum is not unique within eps, or whose inputs are not finite, is
//     re-done by the exact full scan (reference operation order); otherwise the best is the reference's argmin.
//
// Mono co-pol only (the benchmark configuration); uniform finite LUTs only (host checks).
#pragma once
#include "xsw_device.hpp"

namespace xsw {

constexpr int XB = 8;  // pixels per batch (per-lane state: 5 registers per pixel)

template <typename T, typename TO>
__global__ __launch_bounds__(256) void k_invert_exhaustive(DevTables L, KArgs A, int rows_per_chunk)
{
    extern __shared__ __align__(16) double lds_chunk[];  // [rows_per_chunk][phi_pad]
    __shared__ int sh_bin, sh_nbatch;

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long strips_per_line = (A.samples + 63) >> 6, line_groups = (A.lines + 3) >> 2;
    const long long cols_per_xcd = (strips_per_line + 7) >> 3;
    const long long xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const long long col = xcd * cols_per_xcd + j / line_groups;
    const long long line = (j % line_groups) * 4 + wv;
    const bool tile_ok = (j / line_groups < cols_per_xcd) && col < strips_per_line;  // block-uniform
    if (!tile_ok) return;
    const long long smp = col * 64 + lane;
    const bool in = line < A.lines && smp < A.samples;
    const long long i = in ? line * A.samples + smp : 0;
    const double inf = __builtin_inf();

    Pixel P;
    load_pixel<T>(L, A, i, in, P);
    bool pending = (P.flags & F_NEED_CO) != 0;
    const double ah = 0.5 * P.a_re, bh = 0.5 * P.b_eff;
    const double sn = -P.s_co * A.inv_dsig_co;
    const double wh0 = 0.5 * L.w0, whs = 0.5 / L.inv_wstep;
    const double ddp = 2.0 * whs * whs;
    int my_flat = 0;
    bool ambiguous = (P.flags & F_NEED_CO) && !(P.flags & F_CO_FINITE);
    unsigned long long cand = 0;
    const int ncc = (L.n_phi + 63) >> 6;

    for (;;) {
        // next distinct incidence bin of the tile and the number of 8-pixel batches it needs (block-uniform)
        if (threadIdx.x == 0) { sh_bin = 0x7fffffff; sh_nbatch = 0; }
        __syncthreads();
        const int mb = wave_min_i(pending ? P.i_inc : 0x7fffffff);
        if (lane == 0 && mb != 0x7fffffff) atomicMin(&sh_bin, mb);
        __syncthreads();
        const int cur = sh_bin;
        if (cur == 0x7fffffff) break;
        const bool mine = pending && P.i_inc == cur;
        unsigned long long todo = __ballot(mine);
        if (lane == 0) atomicMax(&sh_nbatch, (__popcll(todo) + XB - 1) / XB);
        __syncthreads();
        const int nbatch = sh_nbatch;
        const double *__restrict__ slice = L.co + (size_t)cur * L.n_w * L.phi_pad;

        for (int b = 0; b < nbatch; ++b) {
            // this wave's pixels of the batch: lane numbers pl[q] (wave-uniform), -1 = none
            int pl[XB];
#pragma unroll
            for (int q = 0; q < XB; ++q) {
                pl[q] = todo ? (__ffsll((long long)todo) - 1) : -1;
                if (todo) todo &= todo - 1;
            }
            double best[XB], second[XB];
            int code[XB];
#pragma unroll
            for (int q = 0; q < XB; ++q) { best[q] = inf; second[q] = inf; code[q] = 0; }

            for (int r0 = 0; r0 < L.n_w; r0 += rows_per_chunk) {
                const int rows = min(rows_per_chunk, L.n_w - r0);
                __syncthreads();  // previous chunk fully consumed
                {
                    const double2 *__restrict__ src = (const double2 *)(slice + (size_t)r0 * L.phi_pad);
                    double2 *dst = (double2 *)lds_chunk;
                    const int nvec = rows * L.phi_pad / 2;
                    for (int v = threadIdx.x; v < nvec; v += 256) dst[v] = src[v];
                }
                __syncthreads();
                const double whr = fma((double)r0, whs, wh0);  // w/2 at the chunk's first row
                for (int cc = 0; cc < ncc; ++cc) {
                    const int ip = cc * 64 + lane;
                    const bool ok = ip < L.n_phi;
                    const int ipc = ok ? ip : 0;
                    const double cph = L.cphi[ipc], sph = L.sphi[ipc];
                    const double *colp = lds_chunk + ipc;
#pragma unroll
                    for (int q = 0; q < XB; ++q) {
                        if (pl[q] < 0) continue;  // wave-uniform
                        const double U = 2.0 * (rd_lane_d(ah, pl[q]) * cph + rd_lane_d(bh, pl[q]) * sph);
                        const double usn = rd_lane_d(sn, pl[q]);
                        double pw = ok ? whr * (whr - U) : inf;           // inactive directions score +inf
                        double dp = ok ? whs * (2.0 * whr - U) + whs * whs : 0.0;
                        double bq = best[q], sq = second[q];
                        int cq = code[q];
                        int r = 0;
                        for (; r + 4 <= rows; r += 4) {
                            double v[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] = colp[(r + k) * L.phi_pad];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const double dd = fma(v[k], A.inv_dsig_co, usn);
                                const double J = fma(dd, dd, pw);
                                sq = vmin(sq, vmax(J, bq));
                                const bool lt = J < bq;
                                bq = lt ? J : bq;
                                cq = lt ? (((r0 + r + k) << 3) | cc) : cq;
                                pw += dp;
                                dp += ddp;
                            }
                        }
                        for (; r < rows; ++r) {
                            const double dd = fma(colp[r * L.phi_pad], A.inv_dsig_co, usn);
                            const double J = fma(dd, dd, pw);
                            sq = vmin(sq, vmax(J, bq));
                            const bool lt = J < bq;
                            bq = lt ? J : bq;
                            cq = lt ? (((r0 + r) << 3) | cc) : cq;
                            pw += dp;
                            dp += ddp;
                        }
                        best[q] = bq; second[q] = sq; code[q] = cq;
                    }
                }
            }
            // wave-level argmin, once per pixel of the batch
#pragma unroll
            for (int q = 0; q < XB; ++q) {
                if (pl[q] < 0) continue;
                const double ahq = rd_lane_d(ah, pl[q]), bhq = rd_lane_d(bh, pl[q]);
                const double gmin = wave_min_d(best[q]);
                const double Tthr = gmin + 1e-9 * (1.0 + fabs(gmin) + (ahq * ahq + bhq * bhq));
                const unsigned long long win = __ballot(best[q] <= Tthr);
                const bool amb = !(gmin < inf) || __popcll(win) != 1 || __ballot(second[q] <= Tthr) != 0ULL;
                const int wl = win ? (__ffsll((long long)win) - 1) : 0;
                const int c = rd_lane_i(code[q], wl);
                if (lane == pl[q]) {
                    my_flat = (c >> 3) * L.n_phi + ((c & 7) * 64 + wl);
                    ambiguous = ambiguous || amb;
                }
                cand += (unsigned long long)L.n_w * L.n_phi;
            }
        }
        if (mine) pending = false;
        __syncthreads();  // sh_bin / sh_nbatch are rewritten at the top of the loop
    }

    // settle the rare pixels whose screening minimum was not unique: exact full scan
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
        atomicAdd(&A.stats[0], (unsigned long long)__popcll(__ballot((P.flags & F_NEED_CO) != 0)));
        atomicAdd(&A.stats[1], cand);
        atomicAdd(&A.stats[2], (unsigned long long)n_exact);
    }
    if (in) store_pixel<TO>(L, A, i, P, my_flat, -1);
}

template <typename T, typename TO>
static hipError_t launch_exhaustive(const DevTables &L, const KArgs &A, hipStream_t stream)
{
    if (L.n_phi > 8 * 64 || L.n_w >= (1 << 20)) return hipErrorInvalidValue;  // code = (iw << 3) | direction chunk
#ifndef XSW_EXH_LDS_KB
#define XSW_EXH_LDS_KB 40
#endif
    const size_t lds_budget = (size_t)XSW_EXH_LDS_KB * 1024;  // 40 KB: four workgroups per CU (measured best of 24..128)
    int rows = (int)(lds_budget / ((size_t)L.phi_pad * sizeof(double)));
    if (rows < 1) return hipErrorInvalidValue;
    if (rows > L.n_w) rows = L.n_w;
    rows &= ~3;  // whole groups of four rows
    if (rows < 4) rows = rows < 1 ? 1 : rows;
    const size_t lds = (size_t)rows * L.phi_pad * sizeof(double);
    const long long strips_per_line = (A.samples + 63) / 64, line_groups = (A.lines + 3) / 4;
    const long long nblocks = 8 * ((strips_per_line + 7) / 8) * line_groups;
    if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_invert_exhaustive<T, TO>), dim3((unsigned)nblocks), dim3(256), lds, stream, L, A, rows);
    return hipGetLastError();
}

}  // namespace xsw
