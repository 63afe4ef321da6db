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
    const unsigned long long n_need = (unsigned long long)__popcll(__ballot((P.flags & F_NEED_CO) != 0));
    if (A.stats && lane == 0) {
        atomicAdd(&A.stats[0], n_need);
        atomicAdd(&A.stats[1], cand);
        atomicAdd(&A.stats[2], (unsigned long long)n_exact);
    }
    if (in) store_pixel<TO>(L, A, i, P, my_flat, -1);
}

// ---------------------------------------------------------------------------------------------------------
// Same sweep with FLOAT32 screening (XSW_ALGO_EXHAUSTIVE): the LUT chunk in LDS, the score and the per-lane
// (best, second, code) are float32 -- twice the VALU rate, half the LDS bytes, 3 registers of state per pixel
// (16 pixels per batch).  The decision stays exact:
//   J32 differs from the real score by at most eps32 (bound below: float32 rounding of LUT, s/dsig, U, w/2 and
//   of the three arithmetic steps); a pixel whose float32 minimum is the only candidate within eps32 is done
//   (no other candidate can be the reference's argmin); every other pixel is finished by the float64
//   branch-and-bound search (co_box_search) inside the disc given by its float32 upper bound.
// p = wh*(wh - U) is evaluated directly per candidate (w/2 is a wave-uniform table word): forward differences
// would accumulate float32 error over the 499 steps.
constexpr int XB32 = 16;

template <typename T, typename TO>
__global__ __launch_bounds__(256) void k_invert_exhaustive32(DevTables L, KArgs A, int rows_per_chunk)
{
    extern __shared__ __align__(16) float lds32[];  // [rows_per_chunk][phi_pad]
    __shared__ int sh_bin, sh_nbatch;
    __shared__ float2 row_terms[256];

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long strips_per_line = (A.samples + 63) >> 6, line_groups = (A.lines + 3) >> 2;
    const long long cols_per_xcd = (strips_per_line + 7) >> 3;
    const long long xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const long long col = xcd * cols_per_xcd + j / line_groups;
    const long long line = (j % line_groups) * 4 + wv;
    if (!((j / line_groups < cols_per_xcd) && col < strips_per_line)) return;  // block-uniform
    const long long smp = col * 64 + lane;
    const bool in = line < A.lines && smp < A.samples;
    const long long i = in ? line * A.samples + smp : 0;
    const float inff = __builtin_inff();

    Pixel P;
    load_pixel<T>(L, A, i, in, P);
    bool pending = (P.flags & F_NEED_CO) != 0;
    const bool fin = (P.flags & F_CO_FINITE) != 0;
    const double ah = fin ? 0.5 * P.a_re : 0.0, bh = fin ? 0.5 * P.b_eff : 0.0;
    const double m2 = ah * ah + bh * bh;
    const double sn = fin ? -P.s_co * A.inv_dsig_co : 0.0;
    const float inv32 = (float)A.inv_dsig_co;
    // float32 error of dd = LUT*inv + sn  (LUT and sn rounded to float32, one fused rounding)
    const float e_dd = 3e-7f * (float)((L.co_absmax + fabs(fin ? P.s_co : 0.0)) * fabs(A.inv_dsig_co));
    const float whmax = (float)(0.5 * (L.w0 + (L.n_w - 1) / L.inv_wstep));
    // float32 error of p = wh*(wh - U): U and wh rounded once from float64, two roundings in the product
    const float e_p = 4e-7f * whmax * (whmax + 2.0f * (float)(fabs(ah) + fabs(bh)));
    int my_flat = 0;
    bool need_exact = (P.flags & F_NEED_CO) && !fin;
    bool need_box = false;
    double box_jub = 0.0;
    unsigned long long cand = 0;
    const int ncc = (L.n_phi + 63) >> 6;

    for (;;) {
        if (threadIdx.x == 0) { sh_bin = 0x7fffffff; sh_nbatch = 0; }
        __syncthreads();
        const int mb = wave_min_i(pending ? P.i_inc : 0x7fffffff);
        if (lane == 0 && mb != 0x7fffffff) atomicMin(&sh_bin, mb);
        __syncthreads();
        const int cur = sh_bin;
        if (cur == 0x7fffffff) break;
        const bool mine = pending && P.i_inc == cur;
        unsigned long long todo = __ballot(mine);
        if (lane == 0) atomicMax(&sh_nbatch, (__popcll(todo) + XB32 - 1) / XB32);
        __syncthreads();
        const int nbatch = sh_nbatch;
        const float *__restrict__ slice = L.co32 + (size_t)cur * L.n_w * L.phi_pad;

        for (int b = 0; b < nbatch; ++b) {
            int pl[XB32];
#pragma unroll
            for (int q = 0; q < XB32; ++q) {
                pl[q] = todo ? (__ffsll((long long)todo) - 1) : -1;
                if (todo) todo &= todo - 1;
            }
            float best[XB32], second[XB32], usn[XB32];
            int code[XB32];
#pragma unroll
            for (int q = 0; q < XB32; ++q) {
                best[q] = inff; second[q] = inff; code[q] = 0;
                usn[q] = pl[q] >= 0 ? (float)rd_lane_d(sn, pl[q]) : 0.0f;
            }

            for (int cc = 0; cc < ncc; ++cc) {
                const int ip = cc * 64 + lane;
                const bool ok = ip < L.n_phi;
                const int ipc = ok ? ip : 0;
                const double cph = L.cphi[ipc], sph = L.sphi[ipc];
                // U in float64, rounded once; inactive directions / empty slots: U = -inf => every score +inf
                float U[XB32];
#pragma unroll
                for (int q = 0; q < XB32; ++q)
                    U[q] = (ok && pl[q] >= 0) ? (float)(2.0 * (rd_lane_d(ah, pl[q]) * cph + rd_lane_d(bh, pl[q]) * sph)) : -inff;

                for (int r0 = 0; r0 < L.n_w; r0 += rows_per_chunk) {
                    const int rows = min(rows_per_chunk, L.n_w - r0);
                    __syncthreads();  // previous chunk fully consumed
                    {
                        const float4 *__restrict__ src = (const float4 *)(slice + (size_t)r0 * L.phi_pad);
                        float4 *dst = (float4 *)lds32;
                        const int nvec = rows * L.phi_pad / 4;
                        for (int v = threadIdx.x; v < nvec; v += 256) dst[v] = src[v];
                        if ((int)threadIdx.x < rows) {  // per-row speed terms {(w/2)^2, -(w/2)}
                            const float wh = L.wh32[r0 + threadIdx.x];
                            row_terms[threadIdx.x] = make_float2(wh * wh, -wh);
                        }
                    }
                    __syncthreads();
                    const float *colp = lds32 + ipc;
                    // rows outermost, the 16 pixels of the batch innermost: one LDS read of the LUT word and one
                    // broadcast read of the row terms serve 16 candidates.  Only (best, second) are tracked per
                    // candidate (v_med3_f32 + v_min_f32); WHERE the best sits is tracked per (chunk, direction
                    // chunk) -- "did best improve in here?" -- and the row is recovered afterwards (recover_row).
                    float prev[XB32];
#pragma unroll
                    for (int q = 0; q < XB32; ++q) prev[q] = best[q];
                    for (int r = 0; r < rows; ++r) {
                        const float v = colp[r * L.phi_pad];
                        const float2 rt = row_terms[r];
#pragma unroll
                        for (int q = 0; q < XB32; ++q) {
                            const float dd = fmaf(v, inv32, usn[q]);
                            const float J = fmaf(dd, dd, fmaf(U[q], rt.y, rt.x));   // dd^2 + wh*(wh - U)
                            second[q] = __builtin_amdgcn_fmed3f(best[q], second[q], J);  // two smallest of the three
                            best[q] = vminf(best[q], J);
                        }
                    }
                    const int here = (r0 << 3) | cc;
#pragma unroll
                    for (int q = 0; q < XB32; ++q) code[q] = best[q] < prev[q] ? here : code[q];
                }
            }
            // wave-level argmin + uniqueness within the float32 error bound, once per pixel
#pragma unroll
            for (int q = 0; q < XB32; ++q) {
                if (pl[q] < 0) continue;
                const float gmin = wave_min_f(best[q]);
                const float m2q = (float)rd_lane_d(m2, pl[q]);
                const float edd = __int_as_float(rd_lane_i(__float_as_int(e_dd), pl[q]));
                const float ep = __int_as_float(rd_lane_i(__float_as_int(e_p), pl[q]));
                const float jfull = fmaxf(gmin + m2q, 0.0f);
                // |J32 - J| <= 2|dd| e_dd + e_dd^2 + e_p + 3 ulp32 of the terms, for every candidate that can matter
                const float eps = 2.0f * (2.0f * sqrtf(jfull + 1.0f) * edd + edd * edd + ep + 4e-7f * (1.0f + fabsf(gmin) + 2.0f * jfull));
                const float Tthr = gmin + eps;
                const unsigned long long win = __ballot(best[q] <= Tthr);
                const bool amb = !(gmin < inff) || __popcll(win) != 1 || __ballot(second[q] <= Tthr) != 0ULL;
                const int wl = win ? (__ffsll((long long)win) - 1) : 0;
                const int c = rd_lane_i(code[q], wl);
                // recover the winner's row: re-score its column inside the chunk where the best last improved, one
                // row per lane, with the sweep's arithmetic (same operands, same fused steps => same bits)
                const int rr0 = c >> 3, wip = (c & 7) * 64 + wl;
                int row = -1;
                if (!amb) {
                    const float Uw = (float)(2.0 * (rd_lane_d(ah, pl[q]) * L.cphi[wip] + rd_lane_d(bh, pl[q]) * L.sphi[wip]));
                    const float usnq = (float)rd_lane_d(sn, pl[q]);
                    for (int k0 = 0; k0 < rows_per_chunk && row < 0; k0 += 64) {
                        const int r = rr0 + k0 + lane;
                        const bool okr = (k0 + lane) < rows_per_chunk && r < L.n_w;
                        const int rc = okr ? r : rr0;
                        const float wh = L.wh32[rc];
                        const float dd = fmaf(slice[(size_t)rc * L.phi_pad + wip], inv32, usnq);
                        const float J = fmaf(dd, dd, fmaf(Uw, -wh, wh * wh));
                        const unsigned long long hit = __ballot(okr && J == gmin);
                        if (hit) row = rr0 + k0 + (__ffsll((long long)hit) - 1);
                    }
                }
                if (lane == pl[q]) {
                    my_flat = (row < 0 ? 0 : row) * L.n_phi + wip;
                    need_box = (amb || row < 0) && (gmin < inff);
                    need_exact = need_exact || (amb && !(gmin < inff));
                    box_jub = ((double)Tthr + (double)m2q) * (1.0 + 1e-6) + 1e-6;  // upper bound of the true minimum
                }
                cand += (unsigned long long)L.n_w * L.n_phi;
            }
        }
        if (mine) pending = false;
        __syncthreads();
    }

    // pixels the float32 sweep could not decide: float64 branch-and-bound inside the disc of their upper bound
    unsigned n_box = 0, n_exact = 0;
    unsigned cand2 = 0;
    unsigned long long bx = __ballot(need_box && !need_exact);
    const CoWindow W = box_from_jub(L, P.mag, P.theta, box_jub);
    while (bx) {
        const int p = __ffsll((long long)bx) - 1;
        bx &= bx - 1;
        bool went_exact = false;
        const int flat = co_box_search(L, rd_lane_i(P.i_inc, p), rd_lane_d(P.s_co, p), rd_lane_d(P.a_re, p),
                                       rd_lane_d(P.b_eff, p), rd_lane_i(W.w_lo, p), rd_lane_i(W.w_hi, p),
                                       rd_lane_i(W.ip_lo, p), rd_lane_i(W.ip_hi, p), rd_lane_i(W.geom, p),
                                       rd_lane_i(W.mdiv, p), A.dsig_co, A.inv_dsig_co, lane,
                                       cand2, went_exact, true /* W may be laid out for a 16/32-lane segment */);
        if (lane == p) my_flat = flat;
        n_box++;
        n_exact += went_exact ? 1u : 0u;
    }
    unsigned long long ex = __ballot(need_exact);
    while (ex) {
        const int p = __ffsll((long long)ex) - 1;
        ex &= ex - 1;
        const int flat = exact_scan_co(L, rd_lane_i(P.i_inc, p), rd_lane_d(P.s_co, p), rd_lane_d(P.a_re, p),
                                       rd_lane_d(P.b_eff, p), A.dsig_co, lane);
        if (lane == p) my_flat = flat;
        n_exact++;
    }
    const unsigned long long n_need = (unsigned long long)__popcll(__ballot((P.flags & F_NEED_CO) != 0));
    if (A.stats && lane == 0) {
        atomicAdd(&A.stats[0], n_need);
        atomicAdd(&A.stats[1], cand + cand2);
        atomicAdd(&A.stats[2], (unsigned long long)n_exact);
        atomicAdd(&A.stats[3], (unsigned long long)n_box);  // pixels finished by the float64 box search
    }
    if (in) store_pixel<TO>(L, A, i, P, my_flat, -1);
}

template <typename T, typename TO>
static hipError_t launch_exhaustive(const DevTables &L, const KArgs &A, hipStream_t stream, bool f32_screen)
{
    if (L.n_phi > 8 * 64 || L.n_w >= (1 << 20)) return hipErrorInvalidValue;  // code = (iw << 3) | direction chunk
#ifndef XSW_EXH_LDS_KB
#define XSW_EXH_LDS_KB 40
#endif
    const size_t lds_budget = (size_t)XSW_EXH_LDS_KB * 1024;  // 40 KB: four workgroups per CU (measured best of 24..128)
    const size_t esz = f32_screen ? sizeof(float) : sizeof(double);
    int rows = (int)(lds_budget / ((size_t)L.phi_pad * esz));
    if (rows < 1) return hipErrorInvalidValue;
    if (rows > L.n_w) rows = L.n_w;
    if (f32_screen && rows > 256) rows = 256;  // row_terms[256]
    rows &= ~3;  // whole groups of four rows
    if (rows < 4) rows = rows < 1 ? 1 : rows;
    const size_t lds = (size_t)rows * L.phi_pad * esz;
    const long long strips_per_line = (A.samples + 63) / 64, line_groups = (A.lines + 3) / 4;
    const long long nblocks = 8 * ((strips_per_line + 7) / 8) * line_groups;
    if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if (f32_screen)
        hipLaunchKernelGGL((k_invert_exhaustive32<T, TO>), dim3((unsigned)nblocks), dim3(256), lds, stream, L, A, rows);
    else
        hipLaunchKernelGGL((k_invert_exhaustive<T, TO>), dim3((unsigned)nblocks), dim3(256), lds, stream, L, A, rows);
    return hipGetLastError();
}

}  // namespace xsw
