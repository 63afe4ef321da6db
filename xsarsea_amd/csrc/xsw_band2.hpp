// `k_invert_band2` (round 5): the pixels whose band holds LONG RUNS of rows -- an a-priori wind far from the sigma0 contour, a
// flat stretch of the GMF, a window that reaches past the monotone rows -- handed over by k_invert_band as 48-byte RECORDS
// (BandRec, xsw_band.hpp).  tests/prune_model.py: contour_bound / joint_rows / refined_band_argmin are the executable
// specification of what is new here.
//
// Why these pixels were slow: the three rays of stage 1 look where the a-priori wind points, so J_ub is the score of the
// contour point IN THAT DIRECTION, while the minimum sits where the contour LUT = s comes closest to m, possibly tens of
// degrees away; and window and band bound each cost term by J_ub on its own, which leaves a long thick strip (contour +- d
// inside the disc: 200..1500 candidates) where only a few dozen candidates satisfy both bounds TOGETHER.
//
// One record per lane first (64 pixels side by side, nothing cooperative, perfect lane use):
//   CONTOUR BOUND   the inverse-row table has the contour's row in any direction for one 2-byte read: a coarse scan over the
//                   window's directions (<= XSW_CONTOUR_PROBES, then halvings of the stride around the best) scores the two
//                   rows around each crossing; every probe is a real candidate, so the smallest score is a valid J_ub -- in
//                   practice the minimum itself to a few percent.  The window is recomputed from it.
//   LIVE ARC        per direction of the new window, step B of the joint shrink (below) from the table alone: the directions
//                   in which no row can satisfy both bounds are dropped; what is left is a short arc (5..30 directions where
//                   the window had 60..180), and the pixel is classed by IT.
// Then, STILL one record per lane (LANE SEARCH), per direction of the arc the
//   JOINT SHRINK    J(r) <= J_ub  =>  Jsig(r) <= J_ub - min Jwind over the rows still in question   (B: a narrower band: two
//                                                                                                    table reads)
//                                 =>  Jwind(r) <= J_ub - min Jsig over the rows still in question   (A: a shorter chord:
//                                                                                                    analytic)
//                   min Jwind over a row interval is a clamped parabola; min Jsig over the rows between two table thresholds
//                   follows from the thresholds (rows >= inv[b] have LUT >= t_b, rows < inv[b'] have LUT < t_b').  B, A, B, A;
//                   no LUT read.  The rows past the monotone ones (the TAIL) keep the chord that the direction's tail minimum
//                   (L.tail_min, level 0) leaves of the bound.
// The few rows that survive (a few dozen per pixel in 2..6 directions) are noted as RUNS (direction, first row, rows) in LDS and
// swept by the record's own lane, four candidates in flight; best and second best are the lane's own, so the settle needs no
// cross-lane step.  A record whose runs do not fit (XSW_RUN_CAP runs, XSW_LANE_CAND_MAX candidates: long tails) takes the
// cooperative passes of round 3 instead (a pixel per S-lane segment, K directions per lane, the same shrink per direction, batched
// sweep).  Measured on a-priori x 0.6: the cooperative passes alone spent 411 VALU instructions per record, mostly on the bound
// arithmetic of directions that end up empty -- per lane that arithmetic runs for 64 records at once.  Exactness: a row is dropped only when a lower bound of its score, deflated for
// the rounding of its own arithmetic, exceeds J_ub = (score of a real candidate) (1 + 1e-9) + 1e-9.
// A pixel whose surviving rows are still too many (long tails on the flat top of a saturating GMF) is passed on to
// k_invert_blocks (list C) with the tightened bound's window.
#pragma once
#include "xsw_band.hpp"

namespace xsw {

#ifndef XSW_CONTOUR_PROBES
#define XSW_CONTOUR_PROBES 16
#endif
#ifndef XSW_JOINT_ROUNDS
#define XSW_JOINT_ROUNDS 2
#endif
#ifndef XSW_JOINT_ROUNDS_EASY
#define XSW_JOINT_ROUNDS_EASY 0  // joint-shrink rounds of the waves that are not refined (band and chord of the record's own bound only)
#endif
#ifndef XSW_B2_ROWS_MAX
#define XSW_B2_ROWS_MAX 4096  // rows (candidates) the live arc may hold after step B: beyond, the pixel is k_invert_blocks's (environment XSW_B2_ROWS_MAX)
#endif

struct Band2Slot {  // 56 bytes per pixel in LDS, read by every lane of its segment (broadcast)
    double s, ah, bh, jub;
    int i_inc, rows /* w_lo | w_hi << 16: the monotone part (w_hi < w_lo: none) */, ipn /* first live direction | live directions << 16 */, tail_n;
    int b_lo, b_hi;  // threshold bins of the band s -+ |dsig| sqrt(jub) (bins_margin; b_hi = XSW_INV_BINS: none above)
};

static_assert(sizeof(Band2Slot) == kBand2SlotBytes && kBand2SlotBytes >= 32, "band_wave<ROLE 2> parks four doubles per lane in these slots");

// |m| and its direction in degrees within [phi0, phi0 + 360), as load_pixel forms them (float32 root / arctangent: box_from_jub's margins cover them)
__device__ __forceinline__ void mag_theta(const DevTables &L, double a, double b, double &mag, double &theta)
{
    mag = (double)__builtin_sqrtf((float)(a * a + b * b));
    double th = (double)atan2f((float)b, (float)a) * 57.295779513082320877;
    if (th < L.phi0) th += 360.0;
    theta = th;
}

// BOUND ARITHMETIC IN FLOAT32.  Everything between the bound and a row interval -- budgets, square roots, threshold bins, chord
// ends -- is a BOUND, not a score: it only has to err on the keeping side.  Round 5's first version did it in float64 with libm's
// fmin / fmax (NaN canonicalisation: three instructions each), the IEEE-correct expansion of sqrtf (~15), 64-bit selects and
// conversions: 119 VALU instructions per chord, 166 per table step -- 4 500 per pass, which is why the joint shrink cost more
// than the rows it saved (LABBOOK section 10).  Here: float32 throughout, v_sqrt_f32 itself (1 ulp), v_min / v_max without
// canonicalisation (operands are never NaN), and every result widened by XSW_BOUND_SLACK (relative to the magnitudes that went
// into it: float32 rounding of the inputs and of a handful of operations is ~1e-7 of those) -- ~20 and ~40 instructions.
// tests/prune_model.py carries the same slack.
#ifndef XSW_BOUND_SLACK
#define XSW_BOUND_SLACK 4e-6f
#endif
__device__ __forceinline__ float sqrt_up(float x) { return __builtin_amdgcn_sqrtf(x) * (1.0f + 4e-6f); }  // >= sqrt(x), x >= 0

// Threshold bins of a slice's inverse-row table around [thr_lo, thr_hi] WITH A BIN OF MARGIN instead of stage 1's exact search
// (one bin is ~0.02 dB, a fraction of a row): t_(b_lo) <= thr_lo - width + rounding and t_(b_hi) >= thr_hi + width - rounding, so
// rows below inv[b_lo] lie below thr_lo and rows from inv[b_hi] on lie above thr_hi whatever the last bits of the bin arithmetic
// (float32: the bin coordinate is good to ~2e-4 of a bin).  b_hi = XSW_INV_BINS: no threshold above thr_hi on the grid.
// (tests/prune_model.py: table_bins_margin)
__device__ __forceinline__ void bins_margin(float t0, float inv_width, float thr_lo, float thr_hi, int &b_lo, int &b_hi)
{
    b_lo = max((int)vminf(vmaxf((thr_lo - t0) * inv_width, 0.0f), (float)(XSW_INV_BINS - 1)) - 1, 0);
    b_hi = min(max((int)__builtin_floorf(vminf(vmaxf((thr_hi - t0) * inv_width, -3.0f), (float)XSW_INV_BINS)) + 2, 0), XSW_INV_BINS);  // (0: thr_hi lies below the whole grid)
}

// The same bins WITHOUT margin, by stage 1's search (band_wave): the largest grid threshold <= thr_lo (bin 0 also stands for
// anything below the grid; the checks repeat the table builder's own expression) and the smallest grid threshold > thr_hi.  Once per
// pixel (the band of the pixel's bound): a bin of margin at either end is a row more per direction, a third of a friendly pixel's sweep.
__device__ __forceinline__ void bins_exact(double t0, double width, double inv_width, double thr_lo, double thr_hi, int &b_lo, int &b_hi)
{
    int bin = (int)fmin(fmax((thr_lo - t0) * inv_width, 0.0), (double)(XSW_INV_BINS - 1));
    if (bin > 0 && fma((double)bin, width, t0) > thr_lo) --bin;
    if (bin > 0 && fma((double)bin, width, t0) > thr_lo) bin = 0;
    int bhi = (int)fmin(fmax((thr_hi - t0) * inv_width, -1.0), (double)XSW_INV_BINS) + 1;
    if (bhi < XSW_INV_BINS && !(fma((double)bhi, width, t0) > thr_hi)) ++bhi;
    if (bhi < XSW_INV_BINS && !(fma((double)bhi, width, t0) > thr_hi)) bhi = XSW_INV_BINS;
    b_lo = bin;
    b_hi = bhi;
}

// smallest wind term wh^2 - 2 uh wh + m2 over wh in [wa, wb] (the parabola's minimum clamped into the interval: a lower bound of
// the minimum over the rows in between), deflated for its own rounding
__device__ __forceinline__ float jw_lower(float uh, float m2, float wa, float wb)
{
    const float t = vminf(vmaxf(uh, wa), wb);
    return fmaf(t, t - 2.0f * uh, m2) - XSW_BOUND_SLACK * (m2 + t * (t + 2.0f * fabsf(uh)));
}

// rows with wh^2 - 2 uh wh + m2 <= bud, as index interval [c_lo, c_hi] (inflated; XSW_CHORD_MRG + relative slack in index units);
// false: none
__device__ __forceinline__ bool chord_budget(float uh, float m2, float bud, float wh0, float inv_whs, int &c_lo, int &c_hi)
{
    const float disc = fmaf(uh, uh, bud - m2) + XSW_BOUND_SLACK * (fmaf(uh, uh, m2) + fabsf(bud));
    const float h = sqrt_up(vmaxf(disc, 0.0f));
    const float xc = (uh - wh0) * inv_whs, xh = fmaf(h, inv_whs, (float)XSW_CHORD_MRG + XSW_BOUND_SLACK * fabsf(xc));
    c_lo = (int)__builtin_ceilf(vmaxf(xc - xh, -4.0f));
    c_hi = (int)__builtin_floorf(vminf(xc + xh, 40000.0f));
    return disc >= 0.0f;
}

// CONTOUR BOUND, one pixel per lane: the smallest screening score among the two rows around the crossing LUT = s (inverse-row
// table, bin of s) of <= XSW_CONTOUR_PROBES directions of [ip_lo, ip_lo + ncols), then of the directions half a stride to either
// side of the best, halving.  Rows are clamped into [w_lo, r_top] (the monotone part of the window).  inf: nothing probed.
__device__ __forceinline__ double contour_scan(const DevTables &L, bool on, int i_inc, double s, double ah, double bh, double inv_dsig, int ip_lo,
                                               int ncols, int w_lo, int r_top)
{
    const double inf = __builtin_inf();
    on = on && r_top >= w_lo && ncols >= 1;
    const int ii = on ? i_inc : 0, lo_c = on ? w_lo : 0, top_c = on ? r_top : 0, ip0 = on ? ip_lo : 0;
    const double *g = L.inv_grid + 3 * ii;
    const int bin = (int)fmin(fmax((s - g[0]) * g[2], 0.0), (double)(XSW_INV_BINS - 1));
    const unsigned short *__restrict__ inv = L.inv_rows + mul24_sv((unsigned)L.phi_pad, (unsigned)(ii * XSW_INV_BINS + (on ? bin : 0)));
    const char *__restrict__ base = (const char *)L.co;
    const unsigned rowB = (unsigned)L.phi_pad * 8u, slice0 = mul24_sv(rowB, mul24_sv((unsigned)L.n_w, (unsigned)ii));
    const double sn = on ? -s * inv_dsig : 0.0, wh0 = 0.5 * L.w0, whs = L.wstep_half;
    const double ahc = on ? ah : 0.0, bhc = on ? bh : 0.0;
    // (the two rows of a probe are neighbours: ONE 16-byte read of the TRANSPOSED slice -- a direction's speeds are contiguous there --
    // instead of two reads 1.5 KB apart in the row-major one: the kernel is bound by cache-line accesses, and every lane's probe is a line)
    const unsigned tcol0 = mul24_sv((unsigned)L.n_phi, (unsigned)ii);
    auto pair_of = [&](int ip, int ra, int rb, double &va, double &vb) {
        const int rp = min(ra, L.n_w - 2);
        const double *__restrict__ col = (const double *)((const char *)L.coT + mul24_sv((unsigned)L.w_pad * 8u, tcol0 + (unsigned)ip));
        const double v0 = col[rp], v1 = col[rp + 1];  // (adjacent: one dwordx4)
        va = ra == rp ? v0 : v1;
        vb = rb == rp ? v0 : v1;
    };
    auto rows_of = [&](int ip, int &ra, int &rb) {
        const int r0 = (int)inv[ip];
        ra = min(max(r0 - 1, lo_c), top_c);
        rb = min(max(r0, lo_c), top_c);
    };
    auto score2 = [&](int ip, int ra, int rb, double va, double vb) {
        const double2 cs = ((const double2 *)L.csphi)[ip];
        const double U = 2.0 * (ahc * cs.x + bhc * cs.y);
        const double wa = fma((double)ra, whs, wh0), wb = fma((double)rb, whs, wh0);
        const double da = fma(va, inv_dsig, sn), db = fma(vb, inv_dsig, sn);
        return vmin(fma(da, da, wa * (wa - U)), fma(db, db, wb * (wb - U)));
    };
    const int stride = max(1, (ncols + XSW_CONTOUR_PROBES - 1) / XSW_CONTOUR_PROBES);
    double jc = inf;
    int best = ip0;
    constexpr int NB = 4;  // probes in flight: their table reads, then their LUT reads, then the scores
#pragma unroll 1
    for (int j0 = 0; j0 < XSW_CONTOUR_PROBES; j0 += NB) {
        if (ballot64(on && j0 * stride < ncols) == 0ULL) break;
        int ip[NB], ra[NB], rb[NB];
        double va[NB], vb[NB];
        bool ok[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int off = (j0 + u) * stride;
            ok[u] = on && off < ncols;
            ip[u] = ip0 + (ok[u] ? off : 0);
            rows_of(ip[u], ra[u], rb[u]);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) pair_of(ip[u], ra[u], rb[u], va[u], vb[u]);
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const double j = ok[u] ? score2(ip[u], ra[u], rb[u], va[u], vb[u]) : inf;
            best = j < jc ? ip[u] : best;
            jc = vmin(jc, j);
        }
    }
    int h = on ? stride : 1;
#pragma unroll 1
    while (ballot64(h > 1) != 0ULL) {
        const bool act = h > 1;
        h = act ? (h + 1) >> 1 : h;
        int ip[2], ra[2], rb[2];
        double va[2], vb[2];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = best + (u ? h : -h);
            ok[u] = act && p >= ip0 && p < ip0 + ncols;
            ip[u] = ok[u] ? p : ip0;
            rows_of(ip[u], ra[u], rb[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) pair_of(ip[u], ra[u], rb[u], va[u], vb[u]);
        double j2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) j2[u] = ok[u] ? score2(ip[u], ra[u], rb[u], va[u], vb[u]) : inf;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            best = j2[u] < jc ? ip[u] : best;
            jc = vmin(jc, j2[u]);
        }
    }
    return jc;
}

// The pixel's constants of the bound arithmetic, float32 (the bound itself inflated once more for their rounding)
struct Bound32 {
    float s, uhx, uhy /* ah, bh */, m2, jub, abs_dsig, abs_inv, wh0, whs, inv_whs, t0, width, inv_width;
};
__device__ __forceinline__ Bound32 bound32(const DevTables &L, int i_inc, double s, double ah, double bh, double jub, double dsig)
{
    Bound32 b;
    const double *g = L.inv_grid + 3 * i_inc;
    b.s = (float)s; b.uhx = (float)ah; b.uhy = (float)bh; b.m2 = (float)(ah * ah + bh * bh);
    b.jub = (float)jub * (1.0f + 1e-5f) + 1e-5f;  // (float32 sigma0 and thresholds: ~1e-6 dB, i.e. 1e-5 of a dsig of 0.1 dB, twice that in the score)
    b.abs_dsig = fabsf((float)dsig) * (1.0f + 1e-6f); b.abs_inv = 1.0f / b.abs_dsig;
    b.wh0 = (float)(0.5 * L.w0); b.whs = (float)L.wstep_half; b.inv_whs = (float)(2.0 * L.inv_wstep);
    b.t0 = (float)g[0]; b.width = (float)g[1]; b.inv_width = (float)g[2];
    return b;
}

// Step B of the joint shrink for ONE direction: rows [lo, hi] of the monotone part -> the rows whose LUT value lies within
// +- |dsig| sqrt(J_ub - min Jwind over [lo, hi]) of s, from two reads of the direction's inverse-row column; vlo <= LUT < vhi
// holds for what is left (grid thresholds, widened by their float32 rounding; -inf / +inf: none).  false: nothing is left.
__device__ __forceinline__ bool joint_b(const unsigned short *__restrict__ inv_col /* &inv_rows[slice][0][ip] */, unsigned pitch /* phi_pad */,
                                        const Bound32 &b, float uh, int &lo, int &hi, float &vlo, float &vhi)
{
    const float inf = __builtin_inff();
    const float bud = b.jub - jw_lower(uh, b.m2, fmaf((float)lo, b.whs, b.wh0), fmaf((float)hi, b.whs, b.wh0));
    const float d = fmaf(sqrt_up(vmaxf(bud, 0.0f)), b.abs_dsig, 2e-5f);
    int b_lo, b_hi;
    bins_margin(b.t0, b.inv_width, b.s - d, b.s + d, b_lo, b_hi);
    const int ra = (int)inv_col[mul24_sv(pitch, (unsigned)b_lo)];
    const int rb = (int)inv_col[mul24_sv(pitch, (unsigned)min(b_hi, XSW_INV_BINS - 1))];
    lo = max(lo, ra);
    hi = b_hi < XSW_INV_BINS ? min(hi, rb - 1) : hi;
    vlo = b_lo > 0 ? fmaf((float)b_lo, b.width, b.t0) - 2e-5f : -inf;
    vhi = b_hi < XSW_INV_BINS ? fmaf((float)b_hi, b.width, b.t0) + 2e-5f : inf;
    return bud >= 0.0f && lo <= hi;
}
// smallest sigma0 term ((LUT - s) / dsig)^2 of rows whose LUT value lies in [vlo, vhi), deflated
__device__ __forceinline__ float js_lower(const Bound32 &b, float vlo, float vhi)
{
    const float dmin = vmaxf(0.0f, vmaxf(vlo - b.s, b.s - vhi)) * b.abs_inv * (1.0f - 1e-5f);
    return dmin * dmin;
}

// LIVE ARC, one pixel per lane: first / last direction of [ip_lo, ip_lo + ncols) in which the band of the refined bound (ONE pair of
// threshold bins per pixel: s -+ |dsig| sqrt(J_ub)) holds a row inside the window whose wind term can stay below the bound, or a
// tail row inside the tail's chord -- two table reads and a clamped parabola per direction -- and the rows those directions hold
// (an upper bound of the sweep).  XSW_ARC_UNROLL directions per trip: their reads are in flight together.
#ifndef XSW_ARC_UNROLL
#define XSW_ARC_UNROLL 4
#endif
__device__ __forceinline__ void live_arc(const DevTables &L, bool on, int i_inc, double s, double ah, double bh, double jub, double dsig,
                                         int ip_lo, int ncols, int w_lo, int w_hi, int tail_n, int &first, int &last, int &total_rows, int &max_rows)
{
    // EIGHT directions per trip from ONE 16-byte read of each table row: the kernel is bound by the texture addresser (85 % busy, 557
    // cache-line accesses per record: every lane is another record, so every load touches 64 lines) -- a 2-byte read per direction
    // walked the same two table rows line by line, eight accesses where one does.  The walk starts at the aligned group of eight
    // that holds the window's first direction (a group may reach up to 7 entries past the row's last direction -- the row's pad, the next row, or the 64 bytes of slack every table is allocated with; those entries and the directions outside the window
    // are masked).
    constexpr int G = 8;
    const float inf = __builtin_inff();
    const int ii = on ? i_inc : 0;
    const Bound32 B = bound32(L, ii, s, ah, bh, jub, dsig);
    const double *g = L.inv_grid + 3 * ii;
    const double d = (double)__builtin_sqrtf((float)fmax(jub, 0.0)) * (1.0 + 1e-6) * fabs(dsig) + 1e-9;
    int b_lo, b_hi;
    bins_exact(g[0], g[1], g[2], s - d, s + d, b_lo, b_hi);
    const unsigned short *__restrict__ inv_a = L.inv_rows + mul24_sv((unsigned)L.phi_pad, (unsigned)(ii * XSW_INV_BINS + (on ? b_lo : 0)));
    const unsigned short *__restrict__ inv_b = L.inv_rows + mul24_sv((unsigned)L.phi_pad, (unsigned)(ii * XSW_INV_BINS + (on ? min(b_hi, XSW_INV_BINS - 1) : 0)));
    const bool capped = b_hi < XSW_INV_BINS;
    const double *__restrict__ tmin = L.tail_min ? L.tail_min + mul24_sv((unsigned)L.phi_pad, (unsigned)(ii * (XSW_TAIL_LEVELS + 1))) : nullptr;
    const bool rows_ok = w_hi >= w_lo;
    first = 0x7fffffff; last = -1; total_rows = 0; max_rows = 0;
    const int g0 = on ? (ip_lo & ~(G - 1)) : 0, ip_end = on ? ip_lo + ncols : 0;  // directions [ip_lo, ip_end)
    const int ngroups = wave_max_i(on ? (ip_end - g0 + G - 1) / G : 0);
    const unsigned long long any_tail = ballot64(on && tail_n > 0);
#pragma unroll 1
    for (int k = 0; k < ngroups; ++k) {
        const int gp = g0 + k * G;
        const bool gact = on && gp < ip_end;
        const int gc = gact ? gp : 0;  // (see above: reads past the row's end stay inside the table's allocation)
        const uint4 qa = *(const uint4 *)(inv_a + gc), qb = *(const uint4 *)(inv_b + gc);
        const unsigned wa[4] = {qa.x, qa.y, qa.z, qa.w}, wb[4] = {qb.x, qb.y, qb.z, qb.w};
        float4 c4[G / 2];
#pragma unroll
        for (int u = 0; u < G / 2; ++u) c4[u] = ((const float4 *)((const float2 *)L.csphi32 + gc))[u];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int ip = gc + u;
            const bool act = gact && ip >= ip_lo && ip < ip_end;
            const int ra = (int)((wa[u >> 1] >> ((u & 1) * 16)) & 0xffffu), rb = (int)((wb[u >> 1] >> ((u & 1) * 16)) & 0xffffu);
            const float cx = (u & 1) ? c4[u >> 1].z : c4[u >> 1].x, cy = (u & 1) ? c4[u >> 1].w : c4[u >> 1].y;
            const float uh = B.uhx * cx + B.uhy * cy;
            const int lo = max(w_lo, ra), hi = capped ? min(w_hi, rb - 1) : w_hi;
            const bool fits = act && rows_ok && lo <= hi && !(jw_lower(uh, B.m2, fmaf((float)lo, B.whs, B.wh0), fmaf((float)max(hi, lo), B.whs, B.wh0)) > B.jub);
            int n = fits ? hi - lo + 1 : 0;
            if (any_tail != 0ULL) {  // wave-uniform
                const float tm = tmin ? (float)tmin[act ? ip : 0] - 2e-5f : -inf;  // level 0: the direction's own tail minimum
                int c_lo, c_hi;
                const bool hit = chord_budget(uh, B.m2, B.jub - js_lower(B, tm, inf), B.wh0, B.inv_whs, c_lo, c_hi);
                const int r2 = max(max(w_hi + 1, w_lo), c_lo), l2 = min(w_hi + tail_n, c_hi);
                n += (act && tail_n > 0 && hit && l2 >= r2) ? l2 - r2 + 1 : 0;
            }
            first = (n > 0 && first == 0x7fffffff) ? ip : first;
            last = n > 0 ? ip : last;
            total_rows += n;
            max_rows = max(max_rows, n);
        }
    }
}

// Lane layout of k_invert_band2's window classes (capacities 4, 6, 8, 12, ..., 96, 128 directions as in k_invert_band): S lanes per pixel
// x K directions per lane.  XSW_B2_DEEP: more directions per lane on fewer lanes (K = 4 / 6 where k_invert_band has 2 / 3) -- twice
// the pixels per pass, twice the independent loads per lane: the passes are chains of dependent round trips at 4 waves per SIMD.
#ifndef XSW_B2_DEEP
#define XSW_B2_DEEP 0
#endif
__host__ __device__ constexpr int b2_seg(int c) { return (XSW_B2_DEEP && c >= 2 && c <= 9) ? (1 << (c >> 1)) : (c > 10 ? 64 : (2 << (c >> 1))); }
__host__ __device__ constexpr int b2_dirs(int c) { return (XSW_B2_DEEP && c >= 2 && c <= 9) ? ((c & 1) ? 6 : 4) : ((c & 1) ? 3 : 2); }

// One pass of k_invert_band2: 64 / S pixels, one per S-lane segment, K directions per lane (blocked); per direction the joint
// shrink (XSW_JOINT_ROUNDS x (B, A)) and the tail's chord, then the batched sweep and the settle of co_band_pass.
template <int S, int K>
__device__ __forceinline__ void co_band2_pass(const DevTables &L, double dsig, double inv_dsig, int lane, const Band2Slot *slots /* this wave's [64], sorted by class */,
                                              int *res /* [64], by slot */, int first, int count, unsigned &cand, bool count_on, int rounds /* wave-uniform: joint-shrink rounds (0: band and chord of the pixel's bound only) */)
{
    const double inf = __builtin_inf();
    const int q = lane / S, sl = lane & (S - 1);
    const bool valid = q < count;
    const int owner = valid ? first + q : lane;
    Band2Slot B = slots[owner];
    if (!valid) { B.s = 0.0; B.ah = 0.0; B.bh = 0.0; B.jub = -1.0; B.i_inc = 0; B.rows = 0xffff0000 /* w_lo 0, w_hi -1 */; B.ipn = 0; B.tail_n = 0; B.b_lo = 0; B.b_hi = 0; }
    const int ip_lo = B.ipn & 0xffff, ncols = (int)((unsigned)B.ipn >> 16);
    const int w_lo = B.rows & 0xffff, w_hi = B.rows >> 16, tail_n = B.tail_n;
    const double s = B.s, ah = B.ah, bh = B.bh, m2 = ah * ah + bh * bh, sn = -s * inv_dsig;
    const double wh0 = 0.5 * L.w0, whs = L.wstep_half;
    const Bound32 Q = bound32(L, B.i_inc, s, ah, bh, B.jub, dsig);  // (idle segments: jub = -1: nothing passes)
    const float finf = __builtin_inff();
    const char *__restrict__ base = (const char *)L.co;
    const unsigned rowB = (unsigned)L.phi_pad * 8u;
    const unsigned slice0 = mul24_sv(rowB, mul24_sv((unsigned)L.n_w, (unsigned)B.i_inc));
    const unsigned short *__restrict__ inv_slice = L.inv_rows + mul24_sv((unsigned)L.phi_pad, (unsigned)(B.i_inc * XSW_INV_BINS));
    const double *__restrict__ tmin = L.tail_min ? L.tail_min + mul24_sv((unsigned)L.phi_pad, (unsigned)(B.i_inc * (XSW_TAIL_LEVELS + 1))) : nullptr;
    double best = inf, second = inf;
    int brow = 0, bip = 0;
    bool overflow = false;
    const int nchunks = S == 64 ? (__builtin_amdgcn_readfirstlane(ncols) + 64 * K - 1) / (64 * K) : 1;  // S == 64: one pixel, wave-uniform
#pragma unroll 1
    for (int ch = 0; ch < nchunks; ++ch) {
        bool act[K];
        int ip[K], r[K], nrow[K], n1[K], gap[K];
        unsigned off0[K];
        double U[K];
        int nmax = 0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int vcol = sl + S * j + S * K * ch;
            act[j] = valid && vcol < ncols;
            ip[j] = ip_lo + (act[j] ? vcol : 0);
            const double2 cs = ((const double2 *)L.csphi)[ip[j]];
            U[j] = 2.0 * (ah * cs.x + bh * cs.y);
            const float uh = (float)(0.5 * U[j]);
            off0[j] = slice0 + (unsigned)ip[j] * 8u;
            // round 0: the band of the pixel's bound (one pair of bins per pixel) and the chord of the disc
            int lo, hi;
            bool some;
            {
                const int ra = (int)inv_slice[mul24_sv((unsigned)L.phi_pad, (unsigned)B.b_lo) + (unsigned)ip[j]];
                const int rb = (int)inv_slice[mul24_sv((unsigned)L.phi_pad, (unsigned)min(B.b_hi, XSW_INV_BINS - 1)) + (unsigned)ip[j]];
                int c_lo, c_hi;
                const bool hit = chord_budget(uh, Q.m2, Q.jub, Q.wh0, Q.inv_whs, c_lo, c_hi);
                lo = max(max(w_lo, ra), c_lo);
                hi = min(B.b_hi < XSW_INV_BINS ? min(w_hi, rb - 1) : w_hi, c_hi);
                some = act[j] && hit && hi >= lo;
            }
#pragma unroll 1
            for (int it = 0; it < rounds; ++it) {  // the joint shrink proper: per-direction budgets (ONE copy of the code for every round count: the kernel's size is felt in the instruction cache)
                float vlo, vhi;
                // (a lane that is through keeps harmless bounds: its reads land in the table, its result is discarded)
                int l2 = some ? lo : 0, h2 = some ? hi : 0;
                const bool keep = joint_b(inv_slice + ip[j], (unsigned)L.phi_pad, Q, uh, l2, h2, vlo, vhi);
                int c_lo, c_hi;
                const bool hit = chord_budget(uh, Q.m2, Q.jub - js_lower(Q, vlo, vhi), Q.wh0, Q.inv_whs, c_lo, c_hi);
                some = some && keep && hit;
                lo = max(l2, c_lo);
                hi = min(h2, c_hi);
                some = some && hi >= lo;
            }
            r[j] = some ? lo : 0;
            n1[j] = some ? hi - lo + 1 : 0;
            // the tail: rows w_hi + 1 .. w_hi + tail_n, inside the chord the direction's tail minimum leaves of the bound
            int n2 = 0, r2 = 0;
            if (tail_n > 0) {
                const float tm = tmin ? (float)tmin[ip[j]] - 2e-5f : -finf;
                int c_lo, c_hi;
                const bool hit = chord_budget(uh, Q.m2, Q.jub - js_lower(Q, tm, finf), Q.wh0, Q.inv_whs, c_lo, c_hi);
                r2 = max(max(w_hi + 1, w_lo), c_lo);
                const int l2 = min(w_hi + tail_n, c_hi);
                n2 = (act[j] && hit && l2 >= r2) ? l2 - r2 + 1 : 0;
            }
            if (n1[j] == 0) { r[j] = r2; gap[j] = 0; n1[j] = 0x7fffffff; nrow[j] = n2; }  // tail only: the run starts at its first row
            else { gap[j] = n2 > 0 ? r2 - (r[j] + n1[j]) : 0; nrow[j] = n1[j] + n2; }
            nmax = max(nmax, nrow[j]);
        }
        const int row_top = w_hi + tail_n;  // (clamp of the masked lanes' rows)
#pragma unroll 1
        for (int t0r = 0; t0r < XSW_SWEEP_MAX; t0r += XSW_BAND_BATCH) {
            unsigned long long left[K], any_left = 0ULL;
#pragma unroll
            for (int j = 0; j < K; ++j) { left[j] = ballot64(t0r < nrow[j]); any_left |= left[j]; }
            if (any_left == 0ULL) break;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (left[j] == 0ULL) continue;
                double v[XSW_BAND_BATCH];
                int rc[XSW_BAND_BATCH];
#pragma unroll
                for (int u = 0; u < XSW_BAND_BATCH; ++u) {
                    const int tt = t0r + u;
                    rc[u] = min(max(r[j] + tt + (tt >= n1[j] ? gap[j] : 0), 0), max(row_top, 0));
                    v[u] = ld_co(base, off0[j], rc[u], rowB);
                }
#pragma unroll
                for (int u = 0; u < XSW_BAND_BATCH; ++u) {
                    const bool inb = t0r + u < nrow[j];
                    const double wh = fma((double)rc[u], whs, wh0);
                    const double dd = fma(v[u], inv_dsig, sn);
                    double J = fma(dd, dd, wh * (wh - U[j]));
                    J = inb ? J : inf;
                    second = vmin(second, vmax(J, best));
                    const bool lt = J < best;
                    brow = lt ? rc[u] : brow;
                    bip = lt ? ip[j] : bip;
                    best = vmin(best, J);
                }
            }
            if (count_on) {  // (statistics run: wave-uniform switch)
#pragma unroll
                for (int j = 0; j < K; ++j)
#pragma unroll
                    for (int u = 0; u < XSW_BAND_BATCH; ++u) cand += (unsigned)__popcll(ballot64(t0r + u < nrow[j]));
            }
        }
        overflow = overflow || nmax > XSW_SWEEP_MAX;
    }
    const int bflat = (int)__umul24((unsigned)brow, (unsigned)L.n_phi) + bip;
    const double gmin = S == 64 ? wave_min_d(best) : seg_min_d<S>(best);
    const double T = gmin + 1e-9 * (1.0 + fabs(gmin) + m2);
    const unsigned long long amb = ballot64(valid && (second <= T || overflow)), surv = ballot64(valid && best <= T);
    const unsigned long long segmask = S == 64 ? ~0ULL : (((1ULL << (S & 63)) - 1ULL) << ((q * S) & 63));
    const bool bad = (amb & segmask) != 0ULL || __popcll(surv & segmask) != 1 || !(gmin < 1e300);
    if (valid && ((!bad && best <= T) || (bad && sl == 0))) res[owner] = bad ? -1 : bflat;  // -1: undecided here
}

// REFINE one record per lane: contour bound, the window of the tightened bound, live arc.  The refined record carries the new band
// radius (from which the search recovers the bound), the window's rows, the live arc in place of the window's directions, the
// tail rows the new window still holds; F_TO_C when the pixel is passed on to k_invert_blocks (no arc, or more rows left than
// A.b2_rows_max).
__device__ __forceinline__ BandRec band2_refine(const DevTables &L, const KArgs &A, const BandRec &r, bool searchable, double &jub_out)
{
    const int i_inc = r.inc_tail & 0xffff, tail_old = (int)((unsigned)r.inc_tail >> 16);
    const int w_lo_o = r.rows & 0xffff, w_hi_o = r.rows >> 16;
    const int ip_lo_o = r.ipn & 0xffff, ncols_o = (int)((unsigned)r.ipn >> 16);
    const double s = r.s, ah = r.ah, bh = r.bh, m2 = ah * ah + bh * bh, abs_dsig = fabs(A.dsig_co);
    // the bound stage 1 had (recovered from the band's radius: already inflated), then the contour's
    const double rs = (double)r.d * fabs(A.inv_dsig_co);
    double jub = rs * rs * (1.0 + 1e-12);
    const double jc = contour_scan(L, searchable, i_inc, s, ah, bh, A.inv_dsig_co, ip_lo_o, ncols_o, w_lo_o, w_hi_o);
    if (jc < 1e300) jub = fmin(jub, (jc + m2) * (1.0 + 1e-9) + 1e-9);
    jub_out = jub;
    // the window of the tightened bound, inside the old one (both hold every candidate that can still win)
    double mag, theta;
    mag_theta(L, 2.0 * ah, 2.0 * bh, mag, theta);
    const CoWindow W = box_from_jub(L, mag, theta, searchable ? jub : 0.0);
    const int w_lo = max(W.w_lo, w_lo_o), w_top = min(W.w_hi, w_hi_o + tail_old);
    const int w_hi = min(w_top, w_hi_o), tail_n = tail_old > 0 ? max(w_top - w_hi_o, 0) : 0;
    const int ip_lo = max(W.ip_lo, ip_lo_o), ip_hi = min(W.ip_hi, ip_lo_o + ncols_o - 1);
    int a_first, a_last, rows_total, rows_dir;
    live_arc(L, searchable, i_inc, s, ah, bh, jub, A.dsig_co, ip_lo, ip_hi - ip_lo + 1, w_lo, w_hi, tail_n, a_first, a_last, rows_total, rows_dir);
    // (the bound's own candidate is live, so a searchable record always has an arc; stay safe)
    const bool have = searchable && a_last >= a_first && a_first != 0x7fffffff;
    const bool too_many = have && (rows_total > A.b2_rows_max || rows_dir > XSW_SWEEP_MAX) && A.list_c != nullptr;  // (a direction beyond the sweep's XSW_SWEEP_MAX rows would leave the pixel undecided: list G)  // the block pyramid's (list C)
    BandRec q = r;
    q.inc_tail = i_inc | (tail_n << 16);
    q.rows = (w_lo & 0xffff) | (w_hi << 16);
    q.ipn = have ? (a_first | ((a_last - a_first + 1) << 16)) : 0;
    q.flags = r.flags | ((!have || too_many) ? F_TO_C : 0);
    return q;
}

// One wave's (up to) 64 records: refine, one record per lane; class by the live arc; the cooperative passes; wave_tail (cross-pol phase, hand-over of what is still undecided -- list C for the pixels marked F_TO_C, else list G
// -- and the store).
template <typename T, typename TO, bool CR>
__device__ __forceinline__ void band2_run(const DevTables &L, const KArgs &A, const BandRec &r_in, bool in, bool searchable /* in, and the record is a search (not a pixel to pass on) */,
                                          int lane, Band2Slot *__restrict__ slots, int *__restrict__ res_, long long strip)
{
    constexpr int NC = 11;
    int pos = -1, first[NC] = {}, ncls[NC] = {};
    unsigned cand = 0;
    // The refinement costs every lane of the wave (its loops run as long as the widest window of the 64 records): it runs when
    // enough of the wave's records are marked for it (F_B2_HARD, stage 1 of k_invert_band: long run x wide window, or a tail).
    // On a friendly scene a wave of list B holds two or three such records and takes them as they are -- round 3's search.
    double jub;
    BandRec r = r_in;
    const bool refine_wave = __popcll(ballot64(searchable && (r_in.flags & F_B2_HARD) != 0)) >= A.b2_refine_min ||
                             ballot64(searchable && (r_in.flags & F_B2_CROWD) != 0) != 0ULL;  // (a record beyond XSW_B2_AREA is only here to be refined)
    if (refine_wave) {
        r = band2_refine(L, A, r_in, searchable, jub);
    } else {
        const double rs = (double)r_in.d * fabs(A.inv_dsig_co);
        jub = rs * rs * (1.0 + 1e-12);
    }
    const int flags = (in && !searchable) ? (r.flags | F_TO_C) : r.flags;
    const bool mine = searchable && (r.flags & F_TO_C) == 0;
    const int i_inc = r.inc_tail & 0xffff, tail_n = (int)((unsigned)r.inc_tail >> 16);
    const int w_lo = r.rows & 0xffff, w_hi = r.rows >> 16, a_first = r.ipn & 0xffff, nv = mine ? (int)((unsigned)r.ipn >> 16) : 0;
    const double m2 = r.ah * r.ah + r.bh * r.bh;
    int my_flat = -1;
    const bool big = mine;
    {
        const int p2 = 31 - __clz(max(nv, 2) - 1);
        const int myc = !big ? NC : (nv <= 4 ? 0 : min(2 * p2 - 3 + (nv > (3 << (p2 - 1)) ? 1 : 0), NC - 1));
        int base = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const unsigned long long m = __ballot(myc == c);
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            pos = myc == c ? base + rank : pos;
            first[c] = base;
            ncls[c] = __popcll(m);
            base += ncls[c];
        }
#pragma unroll
        for (int c = 0; c + 1 < NC; ++c) {  // part-filled last passes promoted into the next class (band_wave)
            const int np = 64 / b2_seg(c), npn = 64 / b2_seg(c + 1);
            const int rem = ncls[c] % np;
            const int added = (ncls[c + 1] + rem + npn - 1) / npn - (ncls[c + 1] + npn - 1) / npn;
            if (rem > 0 && added == 0) { ncls[c] -= rem; ncls[c + 1] += rem; first[c + 1] -= rem; }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            first[c] = __builtin_amdgcn_readfirstlane(first[c]);
            ncls[c] = __builtin_amdgcn_readfirstlane(ncls[c]);
        }
        if (big) {
            Band2Slot b;
            b.s = r.s; b.ah = r.ah; b.bh = r.bh; b.jub = jub;
            b.i_inc = i_inc; b.rows = r.rows; b.ipn = r.ipn; b.tail_n = tail_n;
            const double *g = L.inv_grid + 3 * i_inc;
            const double d = (double)__builtin_sqrtf((float)fmax(jub, 0.0)) * (1.0 + 1e-6) * fabs(A.dsig_co) + 1e-9;
            bins_exact(g[0], g[1], g[2], r.s - d, r.s + d, b.b_lo, b.b_hi);
            slots[pos] = b;
            res_[pos] = -1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // (a wave that was not refined sweeps as round 3 did: band and chord of the pixel's own bound, no per-direction budgets --
        // their arithmetic costs more than the rows it saves where the bound is tight already)
        auto run = [&](auto seg, auto kk, int c) {
            constexpr int S = decltype(seg)::value, K = decltype(kk)::value;
            for (int p = 0; p < ncls[c]; p += 64 / S) {
                co_band2_pass<S, K>(L, A.dsig_co, A.inv_dsig_co, lane, slots, res_, first[c] + p, min(64 / S, ncls[c] - p), cand, A.stats != nullptr,
                                    refine_wave ? XSW_JOINT_ROUNDS : XSW_JOINT_ROUNDS_EASY);
            }
        };
#define XSW_B2_RUN(c) run(std::integral_constant<int, b2_seg(c)>{}, std::integral_constant<int, b2_dirs(c)>{}, c)
        XSW_B2_RUN(0); XSW_B2_RUN(1); XSW_B2_RUN(2); XSW_B2_RUN(3); XSW_B2_RUN(4); XSW_B2_RUN(5);
        XSW_B2_RUN(6); XSW_B2_RUN(7); XSW_B2_RUN(8); XSW_B2_RUN(9); XSW_B2_RUN(10);
#undef XSW_B2_RUN
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (big && pos >= 0) my_flat = res_[pos];
    }
    const unsigned long long mine_m = ballot64(mine);
    if (A.stats && A.stats_chain && refine_wave && lane == 0) atomicAdd(&A.stats[7], (unsigned long long)__popcll(mine_m));
    wave_tail<T, TO, CR, true>(L, A, (long long)r.idx, in, lane, flags, my_flat, strip, cand, 4);
}

// (the entry band_wave<ROLE 2> uses: a record built in registers; its LDS is the caller's `slots` block: slots, then the run lists)
template <typename T, typename TO, bool CR>
__device__ __forceinline__ void band2_core(const DevTables &L, const KArgs &A, const BandRec &r, bool in, bool searchable, int lane, Band2Slot *__restrict__ slots,
                                           int *__restrict__ res_, long long strip)
{
    band2_run<T, TO, CR>(L, A, r, in, searchable, lane, slots, res_, strip);
}

// Second kernel of the chain: list B's records, 64 per wave, fixed grid, every wave strides over the list.  A list that
// overflowed is continued in the strip mask (k_invert_band marked the pixels whose record did not fit): stage 1 is redone for
// those (band_wave<ROLE 2> builds the record in registers).  Without records (XSW_NO_RECORDS, tests) list B holds pixel indices.
template <typename T, typename TO, bool CR>
__global__ __launch_bounds__(64 * XSW_BAND_WG_WAVES, XSW_BAND2_WAVES) void k_invert_band2(DevTables L, KArgs A)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ Band2Slot slots[XSW_BAND_WG_WAVES][64];
    __shared__ int res_[XSW_BAND_WG_WAVES][64];
    const long long count = (long long)*A.list_b_count;
    const long long nwaves = (long long)gridDim.x * XSW_BAND_WG_WAVES;
    const long long strips_per_line = (A.samples + 63) >> 6, nstrips = strips_per_line * A.lines;
    if (count > (long long)A.list_b_cap && !A.mask_b) {
        // list B overflowed and there are no strip masks: which pixels k_invert_band meant is unknown, so every strip of the
        // raster is walked: stage 1 is redone for every pixel and only the pixels k_invert_band would hand over are searched
        for (long long c = (long long)blockIdx.x * XSW_BAND_WG_WAVES + wv; c < nstrips; c += nwaves) {  // wave-uniform
            const long long line = c / strips_per_line, smp = (c - line * strips_per_line) * 64 + lane;
            const bool in = smp < A.samples;
            band_wave<T, TO, CR, false, 2>(L, A, line * A.samples + (in ? smp : A.samples - 1), in, lane, (BandSlot *)slots[wv], res_[wv], true, c);
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    const long long nlist = count < (long long)A.list_b_cap ? count : (long long)A.list_b_cap;
    for (long long c = (long long)blockIdx.x * XSW_BAND_WG_WAVES + wv; c * 64 < nlist; c += nwaves) {  // wave-uniform
        const long long k = c * 64 + lane;
        const bool in = k < nlist;
        if (A.rec_b) {
            const BandRec r = ((const BandRec *)A.rec_b)[in ? k : nlist - 1];
            band2_core<T, TO, CR>(L, A, r, in, in, lane, slots[wv], res_[wv], -1);
        } else {
            const long long i = (long long)A.list_b[in ? k : nlist - 1];
            band_wave<T, TO, CR, false, 2>(L, A, i, in, lane, (BandSlot *)slots[wv], res_[wv]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (count > (long long)A.list_b_cap) {
        // the pixels that did not fit into list B are marked in mask_b: the marked pixels of the marked strips, in linear order
        for (long long c = (long long)blockIdx.x * XSW_BAND_WG_WAVES + wv; c < nstrips; c += nwaves) {  // wave-uniform
            const unsigned long long m = A.mask_b[c];  // wave-uniform address
            if (m == 0ULL) continue;
            const long long line = c / strips_per_line, smp = (c - line * strips_per_line) * 64 + lane;
            const bool in = smp < A.samples;
            band_wave<T, TO, CR, false, 2>(L, A, line * A.samples + (in ? smp : A.samples - 1), in && ((m >> lane) & 1ULL) != 0ULL, lane,
                                           (BandSlot *)slots[wv], res_[wv], false, c);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

}  // namespace xsw
