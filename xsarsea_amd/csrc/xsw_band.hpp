// Fast path of the branch-and-bound inversion: `k_invert_band` (round 2) and `k_invert_band2` (round 3).  tests/prune_model.py: band_pruned_argmin is its
// executable specification.
//
// Besides the disc |c - m| <= 2 sqrt(J_ub) of the window (xsw_device.hpp), the sigma0 term bounds the candidates on its
// own: both cost terms are >= 0, so ((LUT - s)/dsig)^2 <= J_ub, i.e. |LUT - s| <= d = |dsig| sqrt(J_ub), is necessary for
// the argmin.  Where the LUT columns are non-decreasing in wind speed over the rows of the window (L.mono_rows, checked at
// upload; CMOD5.N itself turns over beyond 24..40 m/s below 41 deg of incidence), that is ONE row interval per direction
// -- 1..4 candidates instead of the 7..40 rows of the window column: 22 (three-ray bound) instead of 552 candidates per
// pixel on the benchmark scene.
//
// Three kernels make one inversion (xsw.hip: launch_invert).  `k_invert_band` finishes every pixel the band rule decides,
// EXCEPT the pixels whose band holds a long run of rows along the a-priori direction (a pass runs as many trips as its longest
// run: one such pixel holds up every pixel of its pass): those go on list B for `k_invert_band2`, the same wave body on listed
// pixels with the rows swept in batches and clipped to the chord of the search disc (ROLE 2 below).  What neither decides is
// appended to the work list (one atomicAdd per wave) and inverted by `k_invert_list` (the general algorithm: window sweep, exact
// scan, any LUT): window outside the monotone rows, non-finite inputs, near-ties, bands longer than XSW_BAND_MAX rows, cross-pol
// pixels the interval rule cannot decide.  Keeping the rare, register-hungry paths out of k_invert_band is what lets it run at
// 8 waves per SIMD.
// A window that reaches PAST the monotone rows (an a-priori wind above the truth, sigma0 near the GMF's saturation) is still
// the band kernels' in two cases (band_wave, stage 1): no row up there can be in the band (TAIL CUT: L.tail_min, a sparse table
// of per-direction minima of the rows past the monotone ones), or there are few enough of them (TAIL SWEEP: the band rule on
// the monotone part, the rows past it swept in full by k_invert_band2).  A work list that overflows is continued in a STRIP
// MASK (KArgs::mask_g / mask_b: one bit per pixel), so that its consumer takes exactly the pixels meant.
//
// Finding the interval: a monotone column is inverted ONCE, at LUT install (L.inv_rows: for 2048 dB thresholds per incidence
// slice, the first row of every direction at or above the threshold; xsw_lutbuild.hpp).  The largest threshold <= s - d
// gives a row at or below the band's first, the smallest threshold > s + d one past a row at or above its last: two 2-byte
// reads per direction instead of a bisection, and the number of rows to score is known before the sweep starts.
//
// Work decomposition: a workgroup = 4 waves = a tile of 4 lines x 64 samples (as k_invert); lane i loads pixel i, finds its
// incidence bin, the upper bound along the a-priori direction and two neighbours (co_window_lanes), the window and the two
// threshold bins, and parks the pixel's search parameters in an LDS slot -- slots SORTED by window class, so that a pass takes
// the next 64/S slots of its class (no per-pass ranking) and every lane picks its result up once, after the last pass.  The
// wave then takes 64/S pixels per pass, one per S-lane segment; a lane owns K = 2 or 3 directions of one pixel, blocked (sl,
// sl + S, ...: every load of a segment reads contiguous words); classes S*K = 4, 6, 8, 12, ..., 96, 128 directions, wider
// windows loop over chunks in the S = 64 class; on each trip a direction group whose lanes have no rows left is skipped:
//   * the lane reads its directions' first / last candidate rows from the table, clips them to the window, and scores that
//     many rows (every row the tables name is scored: at most one at either end lies outside s - d <= LUT <= s + d, and it is
//     a candidate of the window all the same; scores are formed directly, no forward differences);
//   * segment argmin by DPP; the unique candidate within eps of the minimum is the reference's argmin (same settle rule as
//     co_box_search); the winner lane writes it next to the slot.  Near-ties, several survivors, runs longer than
//     XSW_BAND_MAX -> work list.
// The cross-pol search (dual-pol) runs one pixel per lane: search_cr_scan (table + scan of the admissible interval), or the
// interval rule of search_cr_interval when the cross-pol table is absent.
#pragma once
#include "xsw_device.hpp"

namespace xsw {

#ifndef XSW_BAND_MAX
#define XSW_BAND_MAX 64
#endif
// A lane owns K = 2 or 3 directions of one pixel (a property of the window class: capacities S*K = 4, 6, 8, 12, ..., 96, 128), taken
// BLOCKED (lane sl: directions sl, sl + S, ...), so that every load of a segment reads contiguous table / LUT words.
#ifndef XSW_BAND_RAY_D
#define XSW_BAND_RAY_D 2
#endif
// Hand-over caps of the long-run role.  Compile-time: as runtime arguments (the A/B sweep below ran that way) they cost
// k_invert_band 0.3 ms of 33.7 in stage 1; profiles/sweep_run_caps.sh builds its variants with -D flags instead.
// Measured on the hard scenes of DESIGN 7c
// (64 / 24 / 64 -> 128 / 64 / 128 -> 256 / 256 / 256, Mpx/s): a-priori x 0.3 299 -> 418 -> 415, x 0.6 985 -> 1082 -> 1089,
// x 1.6 755 -> 815 -> 768, x 2.5 159 -> 194 -> 221, incidence 17..33 deg x 1.6 269 -> 377 -> 376; the benchmark scene
// 10 148 -> 10 080 -> 9 928 (a few 1e4 very long runs more for k_invert_band2, while k_invert_list's time there is a latency floor).
#ifndef XSW_LONG_RUN_MAX
#define XSW_LONG_RUN_MAX 128     // rows of band along the a-priori direction beyond which a pixel skips k_invert_band2 (straight to the list)
#endif
#ifndef XSW_LONG_RUN_MAX_CUT
#define XSW_LONG_RUN_MAX_CUT 64  // ... of a window that was cut at the last monotone row (its band lies on the flat top)
#endif
#ifndef XSW_SWEEP_MAX
#define XSW_SWEEP_MAX 128        // rows a direction may hold in k_invert_band2's batched sweep before the pixel is left to k_invert_list
#endif
#ifndef XSW_TAIL_SWEEP
#define XSW_TAIL_SWEEP 256  // rows past the monotone ones a window may hold for k_invert_band2's tail sweep (KArgs::tail_max; 0: off;
                            // environment XSW_TAIL_SWEEP).  Measured (Mpx/s, 0 / 96 / 192 / 400 rows): a-priori x 1.6 1107 / 1186 / 1193 / 1191,
                            // x 2.5 252 / 331 / 394 / 396, incidence 17..33 deg x 1.6 421 / 473 / 570 / 560, 17..25 deg 2460 / 2654 / 2612 / 2651
#endif
#ifndef XSW_B2_HARD_AREA
#define XSW_B2_HARD_AREA 256  // band candidates (run x directions) from which a handed pixel is marked for k_invert_band2's refinement
#endif
#ifndef XSW_BAND_SEEDED
#define XSW_BAND_SEEDED 1  // first ray seeded from the inverse-row table (co_window_lanes)
#endif
#ifndef XSW_BAND_RAYS
#define XSW_BAND_RAYS 3   // rays of the upper bound (co_window_lanes)
#endif
#ifndef XSW_BAND_WG_WAVES
#define XSW_BAND_WG_WAVES 4  // waves (= raster lines) per workgroup; measured 1 / 2 / 4 / 8 / 16: 81.0 / 79.4 / 77.3 / 82.6 / 91.2 ms
#endif
#ifndef XSW_BAND_WAVES
#define XSW_BAND_WAVES 8
#endif
#ifndef XSW_BAND_BATCH_S
#define XSW_BAND_BATCH_S 2  // window classes of S >= this many lanes per pixel sweep their rows in batches (co_band_pass)
#endif
#ifndef XSW_BAND_BATCH
#define XSW_BAND_BATCH 4
#endif
#ifndef XSW_BAND_WAVES_CR
#define XSW_BAND_WAVES_CR 7  // the dual-pol instantiation: 8 / 7 / 6 waves per SIMD measured 59.8 / 58.3 / 58.8 ms at 20000^2 (8 spills 12-20 B per lane)
#endif
#ifndef XSW_BAND2_WAVES
#define XSW_BAND2_WAVES 4  // k_invert_band2: 3 / 4 / 5 / 6 / 8 waves per SIMD measured on the hard scenes -- 4 (128 VGPRs) is the best or within noise of it
#endif

struct BandSlot {  // 64 bytes per pixel, read by every lane of its segment (same address: LDS broadcast)
    double sn, thr_lo, thr_hi, ah, bh, m2;
    int inc_bin /* i_inc | threshold bin of s - d << 16 */, rows /* w_lo | w_hi << 16 */, ipn /* ip_lo | ncols << 16 */;
    int bin_hi /* threshold bin above s + d, or -1 */;
};

struct BandRec {  // what k_invert_band hands to k_invert_band2 per pixel (KArgs::rec_b): 48 bytes, written and read coalesced
    double s, ah, bh;  // sigma0 in dB; the ancillary wind / 2 (b: |b| for a 0..180 deg LUT)
    float d;           // band radius |dsig| sqrt(J_ub) as stage 1 inflated it, rounded up: J_ub is recovered from it
    int inc_tail;      // incidence bin | rows of the window past the monotone ones (the tail) << 16
    int rows;          // w_lo | last row of the monotone part << 16 (the window cut at the slice's last monotone row)
    int ipn;           // ip_lo | directions << 16
    unsigned idx;      // pixel index
    int flags;         // the pixel's class bits
};
struct Band2Slot;
constexpr int kBand2SlotBytes = 56;  // sizeof(Band2Slot) (xsw_band2.hpp asserts it): band_wave<ROLE 2> runs on k_invert_band2's LDS block, whose slots are these
template <typename T, typename TO, bool CR>
__device__ __forceinline__ void band2_core(const DevTables &L, const KArgs &A, const BandRec &r, bool in, bool searchable, int lane, Band2Slot *__restrict__ slots,
                                           int *__restrict__ res_, long long strip);

__device__ __forceinline__ unsigned long long ballot64(bool b) { return __builtin_amdgcn_ballot_w64(b); }

#ifndef XSW_BOUND_SLACK
#define XSW_BOUND_SLACK 4e-6f  // outward slack of the float32 bound arithmetic (xsw_band2.hpp: Bound32)
#endif
// LIVE ARC OF A WIDE WINDOW in stage 1 (round 5), one pixel per lane: first / last direction of [ip_lo, ip_lo + ncols) in which a row of
// the band (rows [inv[bin], inv[bhi]) of the direction's inverse-row column, inside [w_lo, w_hi]) can have its WIND term within
// the bound (the smallest of the parabola wh^2 - 2 uh wh + m2 over those rows, deflated, against jub).  A window is the polar
// bounding box of the disc: with an a-priori wind at 0.3 ... 0.6 of the truth it spans +-40 ... 180 deg although the band only
// meets the disc in a few of those directions -- and a wide window with short runs costs stage 2 a wave-pass of its own (~300
// wave-instructions where a friendly pixel costs 50) whatever the directions hold.  Eight directions per 16-byte read of the two
// table rows (k_invert_band2's live arc without tails and row counts: xsw_band2.hpp).
// Stage 1 of k_invert_band sits on a register-allocation edge (63 VGPRs at 8 waves per SIMD): eight directions per step (16-byte
// reads as in k_invert_band2) put 28 bytes of the kernel's state into scratch and cost the benchmark scene 1.1 ms although its
// waves never take the branch; as a real call (noinline) the kernel ran 46 ms instead of 34.  XSW_ARC_G directions per step.
// Returns first | last << 16 (first > last: no live direction).
#ifndef XSW_ARC_G
#define XSW_ARC_G 8
#endif
__device__ __forceinline__ unsigned window_arc(const unsigned short *__restrict__ inv_rows, const float *__restrict__ csphi32, int phi_pad, float wh0, float whs,
                                               bool on, int i_inc, int bin, int bhi, float uhx, float uhy, float jub, int ip_lo, int ncols, int w_lo, int w_hi)
{
    constexpr int G = XSW_ARC_G;
    static_assert(G == 4 || G == 8, "directions per table read");
    const int ii = on ? i_inc : 0;
    const unsigned short *__restrict__ inv_a = inv_rows + mul24_sv((unsigned)phi_pad, (unsigned)(ii * XSW_INV_BINS + (on ? bin : 0)));
    const unsigned short *__restrict__ inv_b = inv_rows + mul24_sv((unsigned)phi_pad, (unsigned)(ii * XSW_INV_BINS + (on ? min(bhi, XSW_INV_BINS - 1) : 0)));
    const bool capped = bhi < XSW_INV_BINS;
    const float m2 = uhx * uhx + uhy * uhy;
    int first = 0x7fff, last = -1;
    const int g0 = on ? (ip_lo & ~(G - 1)) : 0, ip_end = on ? ip_lo + ncols : 0;
    const int ngroups = wave_max_i(on ? (ip_end - g0 + G - 1) / G : 0);
#pragma unroll 1
    for (int k = 0; k < ngroups; ++k) {
        const int gp = g0 + k * G;
        const bool gact = on && gp < ip_end;
        const int gc = gact ? gp : 0;  // (a group may reach up to 7 entries past the row's last direction: into the row's pad, the next row, or the 64 bytes of slack every table is allocated with -- masked below)
        unsigned wa[G / 2], wb[G / 2];
        if constexpr (G == 8) {
            const uint4 qa = *(const uint4 *)(inv_a + gc), qb = *(const uint4 *)(inv_b + gc);
            wa[0] = qa.x; wa[1] = qa.y; wa[2] = qa.z; wa[3] = qa.w; wb[0] = qb.x; wb[1] = qb.y; wb[2] = qb.z; wb[3] = qb.w;
        } else {
            const uint2 qa = *(const uint2 *)(inv_a + gc), qb = *(const uint2 *)(inv_b + gc);
            wa[0] = qa.x; wa[1] = qa.y; wb[0] = qb.x; wb[1] = qb.y;
        }
#pragma unroll
        for (int h = 0; h < G / 2; ++h) {  // two directions per 16-byte read of the cos / sin table
            const float4 c4 = ((const float4 *)((const float2 *)csphi32 + gc))[h];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ip = gc + 2 * h + e;
                const int ra = (int)((wa[h] >> (e * 16)) & 0xffffu), rb = (int)((wb[h] >> (e * 16)) & 0xffffu);
                const float uh = e ? uhx * c4.z + uhy * c4.w : uhx * c4.x + uhy * c4.y;
                const int lo = max(w_lo, ra), hi = capped ? min(w_hi, rb - 1) : w_hi;
                const float t = vminf(vmaxf(uh, fmaf((float)lo, whs, wh0)), fmaf((float)max(hi, lo), whs, wh0));
                const float jw = fmaf(t, t - 2.0f * uh, m2) - XSW_BOUND_SLACK * (m2 + t * (t + 2.0f * fabsf(uh)));
                const bool live = gact && ip >= ip_lo && ip < ip_end && lo <= hi && !(jw > jub);
                first = (live && first == 0x7fff) ? ip : first;
                last = live ? ip : last;
            }
        }
    }
    return last >= 0 ? (unsigned)first | ((unsigned)last << 16) : 0x00007fffu;
}
__device__ __forceinline__ double ld_co(const char *__restrict__ base, unsigned off0, int row, unsigned rowB)
{
    return *(const double *)(base + (off0 + __umul24((unsigned)row, rowB)));
}

#ifndef XSW_CHORD_MRG
#define XSW_CHORD_MRG 2e-3  // index units of slack on a chord's row bounds (k_invert_band2: chord_budget; as box_from_jub)
#endif
template <int S, int K, bool COUNT>
__device__ __forceinline__ void co_band_pass(const DevTables &L, double inv_dsig, int lane,
                                             const BandSlot *slots /* this wave's [64], sorted by class */, int *res /* [64], by slot */,
                                             int first, int count /* slots [first, first + count) -> segments 0 .. count-1 */, unsigned &cand)
{
    constexpr int sweep_max = XSW_BAND_MAX;  // rows a direction may hold
    const double inf = __builtin_inf();
    const int q = lane / S, sl = lane & (S - 1);
    const bool valid = q < count;
    const int owner = valid ? first + q : lane;  // slot index (idle segment: any slot, its contents are overridden below)
    BandSlot B = slots[owner];
    if (!valid) { B.inc_bin = 0; B.rows = 0; B.ipn = 0; B.bin_hi = -1; }  // idle segment: harmless addresses, nothing scored
    const int B_ip_lo = B.ipn & 0xffff, B_ncols = (int)((unsigned)B.ipn >> 16);
    const int w_lo = B.rows & 0xffff, w_hi = B.rows >> 16;
    const double thr_lo = B.thr_lo, thr_hi = B.thr_hi, sn = B.sn;
    const double wh0 = 0.5 * L.w0, whs = L.wstep_half;
    const char *__restrict__ base = (const char *)L.co;
    const unsigned rowB = (unsigned)L.phi_pad * 8u;
    const int i_inc = B.inc_bin & 0xffff;
    // (24-bit multiplies: full rate, where v_mul_lo_u32 takes four issue slots -- four of them per lane and pass; the operands fit:
    // the band kernels run only when n_inc * n_w and n_inc * XSW_INV_BINS stay below 2^24, xsw.hip: band_mul24)
    const unsigned slice0 = mul24_sv(rowB, mul24_sv((unsigned)L.n_w, (unsigned)i_inc));
    // rows of the inverse table (2-byte entries, one per direction, < 4 GB: xsw.hip): the largest threshold <= s - d gives a
    // row at or below the band's first, the smallest threshold > s + d one past a row at or above its last
    const unsigned short *__restrict__ inv_tab = L.inv_rows;
    const unsigned inv_rowB = (unsigned)L.phi_pad * 2u;
    const unsigned inv0 = mul24_sv(inv_rowB, (unsigned)(i_inc * XSW_INV_BINS) + ((unsigned)B.inc_bin >> 16));
    const unsigned inv1 = mul24_sv(inv_rowB, (unsigned)(i_inc * XSW_INV_BINS) + (unsigned)max(B.bin_hi, 0));
    double best = inf, second = inf;
    int brow = 0, bip = 0;
    unsigned ncand = 0;
    bool overflow = false;
    const int nchunks = S == 64 ? (__builtin_amdgcn_readfirstlane(B_ncols) + 64 * K - 1) / (64 * K) : 1;  // S == 64: one pixel, wave-uniform
#pragma unroll 1
    for (int ch = 0; ch < nchunks; ++ch) {
        bool act[K];
        int ip[K], r[K], nrow[K];
        unsigned off0[K];
        double U[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int vcol = sl + S * j + S * K * ch;  // blocked: the lanes of a segment read contiguous directions in every load
            act[j] = valid && vcol < B_ncols;
            ip[j] = B_ip_lo + (act[j] ? vcol : 0);
            const unsigned ipB = (unsigned)ip[j] * 8u;
            const double2 cs = *(const double2 *)((const char *)L.csphi + 2u * ipB);
            U[j] = 2.0 * (B.ah * cs.x + B.bh * cs.y);
            off0[j] = slice0 + ipB;
        }
        // rows to look at: from the tabulated row at or below the band's first to the one at or above its last, inside the window
        int nmax = 0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const unsigned o_first = inv0 + (unsigned)ip[j] * 2u, o_last = inv1 + (unsigned)ip[j] * 2u;
            const int ra = (int)*(const unsigned short *)((const char *)inv_tab + o_first);
            const int rb = (int)*(const unsigned short *)((const char *)inv_tab + o_last);
            r[j] = max(ra, w_lo);
            const int last = B.bin_hi >= 0 ? min(rb - 1, w_hi) : w_hi;
            nrow[j] = act[j] ? last - r[j] + 1 : 0;
            nmax = max(nmax, nrow[j]);
        }
#pragma unroll 1
        for (int t = 0; t < XSW_BAND_MAX; ++t) {  // scalar trip counter; the loop leaves as soon as no lane has rows left
            unsigned long long left[K], any_left = 0ULL;
#pragma unroll
            for (int j = 0; j < K; ++j) { left[j] = ballot64(t < nrow[j]); any_left |= left[j]; }
            if (any_left == 0ULL) break;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (left[j] == 0ULL) continue;  // wave-uniform: the j-th directions of this pass have no rows left (often the upper half)
                // (a lane past its run -- masked below -- may read past the window, up to XSW_BAND_MAX rows: the table is padded by
                // 260 rows, xsw.hip; r[j] <= n_w)
                const int rc = min(r[j] + t, w_hi);  // (unclamped -- the table is padded -- measured slower: masked lanes then touch new cache lines)
                const double v = ld_co(base, off0[j], rc, rowB);
                // the end rows may lie just outside the band: candidates of the window all the same, so scoring them is harmless
                // and cheaper than the two comparisons that would mask them (band kernel at 20000^2: 35.65 -> 33.9 ms)
                const bool inb = t < nrow[j];
                const double wh = fma((double)rc, whs, wh0);
                const double dd = fma(v, inv_dsig, sn);
                double J = fma(dd, dd, wh * (wh - U[j]));
                // a masked lane scores ~9e307 (finite, above the 1e300 "nothing scored" mark): one select on the high word
                J = __hiloint2double(inb ? __double2hiint(J) : 0x7FE00000, __double2loint(J));
                second = vmin(second, vmax(J, best));
                const bool lt = J < best;
                brow = lt ? rc : brow;
                if (K > 1 || S == 64) bip = lt ? ip[j] : bip;  // S == 64: the best may sit in an earlier chunk
                best = vmin(best, J);
                if (COUNT) ncand += inb ? 1u : 0u;
            }
        }
        const bool any = nmax > sweep_max;
        overflow = overflow || any;  // rows left after XSW_BAND_MAX trips
        if (K == 1 && S != 64) bip = ip[0];
    }
    const int bflat = (int)__umul24((unsigned)brow, (unsigned)L.n_phi) + bip;
    const double gmin = S == 64 ? wave_min_d(best) : seg_min_d<S>(best);
    const double T = gmin + 1e-9 * (1.0 + fabs(gmin) + slots[owner].m2);  // re-read: not kept live through the sweep
    const unsigned long long amb = ballot64(valid && (second <= T || overflow)), surv = ballot64(valid && best <= T);
    const unsigned long long segmask = S == 64 ? ~0ULL : (((1ULL << (S & 63)) - 1ULL) << ((q * S) & 63));
    const bool bad = (amb & segmask) != 0ULL || __popcll(surv & segmask) != 1 || !(gmin < 1e300);
    if (valid && ((!bad && best <= T) || (bad && sl == 0))) res[owner] = bad ? -1 : bflat;  // -1: undecided here, left to k_invert_list
    if (COUNT) {
        unsigned c = ncand;
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
        cand += c;
    }
}

// Compact append of the wave's flagged pixels to a work list: one atomicAdd per wave.  Past the capacity the counter keeps
// running (the consumer sees count > cap and falls back to inverting every tile: k_invert_list).
// Returns true for a lane whose pixel did not fit (the caller marks it in the strip mask instead).
__device__ __forceinline__ bool list_append(unsigned *__restrict__ count, unsigned *__restrict__ list, unsigned cap, bool flag, long long i, int lane)
{
    const unsigned long long um = __ballot(flag);
    if (!um) return false;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(count, (unsigned)__popcll(um));
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    const unsigned at = base + __builtin_amdgcn_mbcnt_hi((unsigned)(um >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)um, 0u));
    if (flag && at < cap) list[at] = (unsigned)i;
    return flag && at >= cap;
}
// Marks pixel i (lanes with `flag`) in a strip mask: one atomicOr per wave when the wave's pixels are strip `strip` of the raster
// (lane = sample), one per pixel when they are listed pixels.
__device__ __forceinline__ void mask_mark(unsigned long long *__restrict__ mask, bool flag, long long i, long long strip, long long samples, int lane)
{
    const unsigned long long m = __ballot(flag);
    if (!m) return;
    if (strip >= 0) {
        if (lane == 0) atomicOr(&mask[strip], m);
    } else if (flag) {
        const long long ln = i / samples, sm = i - ln * samples;
        atomicOr(&mask[ln * ((samples + 63) >> 6) + (sm >> 6)], 1ULL << (sm & 63));
    }
}

// What every wave of the band kernels (and of k_invert_blocks) ends with, one pixel per lane: the cross-pol search
// (windspeed.py:252-269; interval rule only, host: L.cr_monotone), the hand-over of the pixels that are still undecided -- to_b:
// list B (k_invert_band2), to_c: list C (k_invert_blocks), everything else and whatever did not fit: list G (k_invert_list), or the
// list's strip mask once it is full -- and the store of the decided ones.
// flags / my_flat: the lane's pixel class and winning flat index (-1: no co-pol answer yet).  strip: the wave's pixels are strip
// `strip` of the raster (lane = sample); -1: listed pixels.
template <typename T, typename TO, bool CR, bool COUNT>
__device__ __forceinline__ void wave_tail(const DevTables &L, const KArgs &A, long long i, bool in, int lane, int flags, int my_flat, long long strip,
                                          unsigned cand, int chain_slot = 0 /* chain statistics: the kernel's own counter (A.stats[chain_slot]) */)
{
    const bool to_b = A.list_b != nullptr && (flags & F_TO_B) != 0, to_c = (flags & F_TO_C) != 0;  // (they ride in `flags`: no register of their own through the passes)
    const double nan = __builtin_nan("");
    int my_icr = -1;
    const bool need_co = (flags & F_NEED_CO) != 0;
    bool unresolved = in && need_co && my_flat < 0;  // not eligible, or undecided by its pass

    // ---- cross-pol search (windspeed.py:252-269), one pixel per lane, interval rule only (host: L.cr_monotone)
    bool need_cr = false;
    if (CR && A.s_cr) {
        const double inc = ld<T>(A.inc, i);
        const T x = ((const T *)A.s_cr)[i];
        const double s_cr = to_db(x, A.is_db);
        const double dsig = A.dsig_cr ? (double)((const T *)A.dsig_cr)[i] : (double)(T)(x * (T)0 + (T)A.dsig_cr_scalar);
        need_cr = in && !(flags & (F_EARLY_NAN | F_CR_RAW_NAN)) && s_cr == s_cr && dsig == dsig;
        const bool here = need_cr && !unresolved;
        const int i_inc_cr = here ? nearest_index(L.inc_cr, L.n_inc_cr, inc, L.inc_cr_uniform != 0, L.inc_cr0, L.inv_inccrstep) : 0;
        const double aco = (here && need_co) ? L.abs_co[my_flat] : nan;  // np.abs(wind_co), table [n_w][n_phi]
        bool undecided = here;
        const bool done = L.inv_cr ? search_cr_scan(L, here, i_inc_cr, s_cr, dsig, need_co, aco, my_icr, undecided)
                                   : search_cr_interval(L, here, i_inc_cr, s_cr, dsig, need_co, aco, my_icr, undecided);
        unresolved = unresolved || (need_cr && (undecided || !done));
        if (need_cr) flags |= F_NEED_CR;
    }
    if (COUNT && A.stats) {
        const unsigned long long done_co = __ballot(in && need_co && my_flat >= 0), done_cr = __ballot(need_cr && !unresolved);
        if (lane == 0) {
            atomicAdd(&A.stats[0], (unsigned long long)__popcll(done_co));
            atomicAdd(&A.stats[1], (unsigned long long)cand);
            if (chain_slot && A.stats_chain) atomicAdd(&A.stats[chain_slot], (unsigned long long)cand);
            atomicAdd(&A.stats[3], (unsigned long long)__popcll(done_cr));
        }
    }
    // hand the undecided pixels over (a pixel that does not fit into its list goes on list G; one that does not fit there either
    // is marked in the list's strip mask -- KArgs::mask_g / mask_b, zeroed before every launch, touched only when a list overflows --
    // and the consumer takes the list, then the marked pixels)
    bool drop_b = false, drop_c = false;
    // (with records, a pixel that is still list B's here is one whose record did not fit -- counted already: straight to the mask)
    if (A.list_b) drop_b = A.rec_b ? (unresolved && to_b) : list_append(A.list_b_count, A.list_b, A.list_b_cap, unresolved && to_b, i, lane);
    if (A.list_c) drop_c = list_append(A.list_c_count, A.list_c, A.list_c_cap, unresolved && to_c && !to_b, i, lane);
    const bool for_c = A.list_c != nullptr && to_c && !to_b && !drop_c;
    const bool drop_g = list_append(A.list_count, A.list, A.list_cap, unresolved && !to_b && !for_c, i, lane);
    if (A.mask_g) {
        if (A.list_b) mask_mark(A.mask_b, drop_b, i, strip, A.samples, lane);
        mask_mark(A.mask_g, drop_g, i, strip, A.samples, lane);
    }
    if (in) {
        if (!unresolved) {
            Pixel Q;  // what store_pixel reads: flags and the ancillary wind (reloaded: not kept live through the passes)
            Q.flags = flags;
            Q.a_re = nan; Q.a_im = nan;
            if (A.anc) {
                typename Cx<T>::type z = ((const typename Cx<T>::type *)A.anc)[i];
                Q.a_re = (double)z.x;
                Q.a_im = (double)z.y;
            }
            store_pixel<TO, CR>(L, A, i, Q, my_flat, my_icr);
        }
    }
}

// One wave's 64 pixels (lane l: pixel i, `in` = the lane has one) through stage 1, the band passes, the cross-pol phase and the
// store.  Body of k_invert_band (tiles of the raster) and k_invert_band2 (pixels of a work list).
// COUNT = true: the statistics instantiation (xsw_stats_enable): candidates are counted per pass; its own kernel so that the
// production kernel carries one copy of each pass (half the code).
// ROLE: 0 = every window class in this kernel (the statistics instantiation; XSW_LONG_RUN=0);
//       1 = pixels whose band holds A.long_run or more rows along the a-priori direction are handed to the second band kernel
//           (list B), the others are swept here (k_invert_band's default);
//       2 = the rows swept in batches and clipped to the disc's chord (k_invert_band2).
template <typename T, typename TO, bool CR, bool COUNT, int ROLE = 0>
__device__ __forceinline__ void band_wave(const DevTables &L, const KArgs &A, long long i, bool in, int lane, BandSlot *__restrict__ slots,
                                          int *__restrict__ res_, bool strip_walk = false /* ROLE 2 walking every strip: the short-run pixels are k_invert_band's */,
                                          long long strip = -1 /* the wave's pixels are strip `strip` of the raster (lane = sample); -1: listed pixels */)
{
    const double nan = __builtin_nan("");
    int flags, my_flat = -1, my_icr = -1;
    unsigned cand = 0;
    constexpr int NC = 11;  // window classes: S lanes x K directions = 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128 (and wider: chunks)
    int pos = -1, first[NC] = {}, ncls[NC] = {};  // slot of this lane's pixel; slot range of each class
    bool skip = false;        // ROLE 2 walking every strip (list B overflowed): not one of the long-run pixels this kernel is for
    // (F_TO_B in `flags`: ROLE 1, left to the second band kernel -- its band holds a long run of rows; F_TO_C: left to
    // k_invert_blocks -- a finite pixel the band rule is not for: its window leaves the monotone rows, or its band holds more rows /
    // candidates than k_invert_band2 takes)
    {
        // ---- stage 1, one pixel per lane: classify, incidence bin, upper bound along the a-priori direction, window
        Pixel P;
        load_pixel<T, false>(L, A, i, in, P);
        flags = P.flags;
        if (CR && A.s_cr) {  // touch the cross-pol inputs now: by the time the passes are over their lines sit in L2, not in HBM
            const T x = ((const T *)A.s_cr)[i];
            const T dr = A.dsig_cr ? ((const T *)A.dsig_cr)[i] : (T)0;
            if (x != x || dr != dr) flags |= F_CR_RAW_NAN;
        }
        const unsigned long long todo = __ballot((P.flags & F_NEED_CO) != 0);
        bool eligb = false;
        int ncols_p = 0;
        if (todo) {
            bool loose = false;
            CoWindow W = co_window_lanes<XSW_BAND_RAYS, XSW_BAND_RAY_D, XSW_BAND_SEEDED != 0>(L, P, A.inv_dsig_co, fabs(A.dsig_co), loose);
            // A window that reaches past the slice's monotone rows is still the band rule's if no row up there can be in the band:
            // every LUT value of the rows >= mono_rows in the window's directions lies above s + d (L.tail_min, a sparse table over
            // the directions: CMOD5.N saturates and then falls back slowly, so this is the common case of an a-priori wind well
            // above the one sigma0 points to; upwind and crosswind saturate several dB apart, hence per direction).  Those rows
            // cannot hold the argmin (their sigma0 term alone exceeds J_ub): the window is cut at the last monotone row.
            int w_hi_e = W.w_hi;
            bool has_tail = false;  // the rows w_hi_e + 1 .. W.w_hi are the window's tail (below)
            if (L.tail_min) {
                const bool fin1 = (P.flags & F_NEED_CO) != 0 && (P.flags & F_CO_FINITE) != 0;
                const int mono1 = L.mono_rows[fin1 ? P.i_inc : 0];
                if (fin1 && W.w_hi >= mono1 && W.ip_hi >= W.ip_lo) {
                    const int len = W.ip_hi - W.ip_lo + 1, k = min(31 - __clz(len), XSW_TAIL_LEVELS);
                    const double *tk = L.tail_min + mul24_sv((unsigned)L.phi_pad, (unsigned)(P.i_inc * (XSW_TAIL_LEVELS + 1) + k));
                    const double lo = fmin(tk[W.ip_lo], tk[k < XSW_TAIL_LEVELS ? W.ip_hi - (1 << k) + 1 : W.ip_lo]);
                    if (P.s_co + W.band_d < lo) w_hi_e = mono1 - 1;
                    // TAIL SWEEP (ROLE 1 / 2): the band does reach up there -- the flat top of a saturating GMF under speckle: the
                    // pixels that used to cost k_invert_list the most (whole windows of 1e4 candidates, one pixel at a time).  The
                    // band rule still holds on the monotone part; the rows past it, at most A.tail_max, are swept in full by
                    // k_invert_band2 (clipped to the disc's chord like every row there): the window is cut and the tail noted.
                    else if (ROLE != 0 && mono1 >= 1 && W.w_hi - mono1 + 1 <= A.tail_max) { has_tail = true; w_hi_e = mono1 - 1; }
                }
            }
            const int nrows_p = w_hi_e - W.w_lo + 1;
            ncols_p = W.ip_hi - W.ip_lo + 1;
            const bool need = (P.flags & F_NEED_CO) != 0 && (P.flags & F_CO_FINITE) != 0 && ncols_p >= 1 && (nrows_p >= 1 || has_tail);
            if (ROLE == 2 && strip_walk) skip = !need || !(w_hi_e < L.mono_rows[need ? P.i_inc : 0]);  // everything k_invert_band did not hand over
            eligb = need && w_hi_e < L.mono_rows[need ? P.i_inc : 0];  // the window stays inside the monotone rows
            if (ROLE != 2 && need && !eligb) flags |= F_TO_C;
            if (COUNT) cand += (unsigned)__popcll(__ballot(eligb)) * (unsigned)(2 * (32 - __clz((L.n_w + 1) >> 1)) + (XSW_BAND_RAYS - 1) * 2 * XSW_RAY_SIDE_STEPS);
            // class of a window by its number of (virtual) columns n: the smallest of 4, 6, 8, 12, ..., 96, 128 that holds it, i.e. S lanes
            // x K directions per lane with K = 2 (capacity 2S) or 3 (capacity 3S, S half as large: twice the pixels per pass of
            // the next power of two).  The slots are written SORTED by class (slot = pixels of narrower classes + rank inside the
            // class): a pass takes the next 64/S slots of its class, no per-pass ranking, and a lane picks its result up from its
            // slot once, after the last pass
            // threshold bins of the slice's inverse table: the largest grid threshold <= s - d (bin 0 also stands for anything
            // below the grid; the check repeats the builder's own expression, k_inv_rows), and the smallest grid threshold
            // > s + d (none: the band may reach the window's last row)
            double thr_lo = P.s_co - W.band_d, thr_hi = P.s_co + W.band_d;
            int bin = 0, bhi = XSW_INV_BINS;
            if (eligb) {
                const double *g = L.inv_grid + 3 * P.i_inc;
                const double t0 = g[0], width = g[1];
                bin = (int)fmin(fmax((thr_lo - t0) * g[2], 0.0), (double)(XSW_INV_BINS - 1));
                if (bin > 0 && fma((double)bin, width, t0) > thr_lo) --bin;
                if (bin > 0 && fma((double)bin, width, t0) > thr_lo) bin = 0;
                bhi = (int)fmin(fmax((thr_hi - t0) * g[2], -1.0), (double)XSW_INV_BINS) + 1;
                if (bhi < XSW_INV_BINS && !(fma((double)bhi, width, t0) > thr_hi)) ++bhi;
                if (bhi < XSW_INV_BINS && !(fma((double)bhi, width, t0) > thr_hi)) bhi = XSW_INV_BINS;
            }
            if (ROLE != 0 && A.arc_min < 0x7fffffff) {
                // WIDE windows narrowed to their live arc before they are classed (window_arc) -- when enough of the wave's pixels are wide
                // to pay for the walk (every lane waits for the widest window: 23 groups of 8 directions for 181)
                // (a window with a tail keeps its directions: the table says nothing about the rows past the monotone ones -- k_invert_band2's
                // live arc, which knows the tail's chord, narrows it)
                const bool widep = eligb && !has_tail && ncols_p >= A.arc_min;
                if (__popcll(__ballot(widep)) >= A.arc_crowd) {
                    const double rs = W.band_d * fabs(A.inv_dsig_co);
                    const float jub32 = (float)(rs * rs) * (1.0f + 1e-5f) + 1e-5f;
                    // (stage 1 holds 63 live VGPRs here: the walk's own state does not fit beside them, and a compiler spill costs the
                    // kernel scratch -- four doubles wait in the lane's own, still unused LDS slot instead)
                    // (ROLE 2 runs on k_invert_band2's LDS block: 64 slots of kBand2SlotBytes, not of sizeof(BandSlot))
                    double *park;
                    if constexpr (ROLE == 2) park = (double *)((char *)slots + lane * kBand2SlotBytes);
                    else park = (double *)&slots[lane];
                    park[0] = thr_lo; park[1] = thr_hi; park[2] = W.band_d; park[3] = P.s_co;
                    __asm__ volatile("" ::: "memory");
                    const unsigned arc = window_arc(L.inv_rows, (const float *)L.csphi32, L.phi_pad, (float)(0.5 * L.w0), (float)L.wstep_half, widep, P.i_inc, bin, bhi,
                                                    (float)(0.5 * P.a_re), (float)(0.5 * P.b_eff), jub32, W.ip_lo, ncols_p, W.w_lo, w_hi_e);
                    __asm__ volatile("" ::: "memory");
                    thr_lo = park[0]; thr_hi = park[1]; W.band_d = park[2]; P.s_co = park[3];
                    const int a_first = (int)(arc & 0xffffu), a_last = (int)(arc >> 16);
                    if (widep && a_last >= a_first) {  // (the bound's own candidate is live: there always is an arc; stay safe)
                        W.ip_lo = a_first;
                        ncols_p = a_last - a_first + 1;
                    }
                }
            }
            bool hard = false;  // ROLE 1: the handed pixel is worth k_invert_band2's refinement (long run x wide window, or a tail)
            int myc = NC;
            if (eligb) {
                const int nv = ncols_p;
                const int p2 = 31 - __clz(max(nv, 2) - 1);  // 2^p2 < n <= 2^(p2 + 1)
                myc = nv <= 4 ? 0 : min(2 * p2 - 3 + (nv > (3 << (p2 - 1)) ? 1 : 0), NC - 1);
            }
            if (ROLE == 1 || (ROLE == 2 && strip_walk)) {
                // LONG runs: the rows the band holds along the a-priori direction (the inverse table's own answer) are what a
                // pass's trip count follows, and one pixel with a long run holds up every pixel of its pass.  Pixels with at least
                // A.long_run such rows go to k_invert_band2, which sees long runs only (batched sweeps, chord clip, fewer waves
                // per SIMD); its strip walk repeats this per-pixel decision.
                int run = 0;
                if (eligb) {
                    // (32-bit element offsets by 24-bit multiplies: band_mul24, the table is < 4 GB)
                    const unsigned tab0 = (unsigned)(P.i_inc * XSW_INV_BINS);
                    auto run_at = [&](int ip) {
                        const int ra = (int)L.inv_rows[mul24_sv((unsigned)L.phi_pad, tab0 + (unsigned)bin) + (unsigned)ip];
                        const int rb = bhi < XSW_INV_BINS ? (int)L.inv_rows[mul24_sv((unsigned)L.phi_pad, tab0 + (unsigned)bhi) + (unsigned)ip] : w_hi_e + 1;
                        return min(rb - 1, w_hi_e) - max(ra, W.w_lo) + 1;
                    };
                    run = run_at(P.ipr);  // (the window's first and last directions as well: hands over 3.5x the pixels for 2 ms less here, 4 ms more there)
                    if (has_tail) run = max(run, 0) + (W.w_hi - w_hi_e);  // the tail's rows count as run
                }
                // (a run beyond XSW_LONG_RUN_MAX rows -- the flat top of a saturating GMF -- would overflow k_invert_band2's sweep
                // after costing it the most: such a pixel goes straight to the general kernel)
                // (without the cap on tail pixels: a-priori x 1.6 1189 -> 963 Mpx/s, x 2.5 377 -> 204: k_invert_band2 drowns in them)
                // (and a WIDE window with a long run -- sigma0 far above what any wind near the a-priori one explains: every direction
                // searched, bands on the saturated top -- costs k_invert_band2 run x chunks of 128 directions trips: beyond
                // A.area_max band candidates (run x directions) the general kernel's block pyramid is cheaper)
                // (round 5: ... unless the wave's strip is CROWDED with such pixels -- A.b2_crowd or more of its 64: a scene whose a-priori
                // wind is far from the sigma0 contour everywhere, not a ship or a rain cell.  Their records reach k_invert_band2 side by
                // side, its waves run the refinement (contour bound, live arc), and a record then costs 0.6 ns where the pyramid costs
                // 1.9: a-priori x 0.3 944 -> 1115 Mpx/s.  A lone such pixel in a wave that is not refined would be swept as it is.)
                // (round 5, measured: handing the long runs to k_invert_band2 marked for its refinement moves 12 % of the pixels of the x 2.5 scene
                // there and nearly all of them come back as too many rows -- 49 + 57 ms where 41 + 63 were; they are the pyramid's)
                const bool long_one = eligb && run > ((w_hi_e < W.w_hi && !has_tail) ? XSW_LONG_RUN_MAX_CUT : XSW_LONG_RUN_MAX);
                const bool big_area = eligb && !long_one && run * ncols_p > A.area_max;
                const bool crowded = __popcll(__ballot(big_area)) >= A.b2_crowd;
                if (big_area && crowded && run * ncols_p <= A.area_crowd_max) flags |= F_B2_CROWD;
                if (long_one || (big_area && (flags & F_B2_CROWD) == 0)) {
                    myc = NC;
                    eligb = false;
                    if (ROLE == 2) skip = true;
                    if (ROLE == 1) flags |= F_TO_C;
                }
                hard = has_tail || run * ncols_p >= XSW_B2_HARD_AREA || ncols_p >= 64;
                const bool handed = eligb && (run >= A.long_run || has_tail || ncols_p >= A.wide_min);  // (a tail is k_invert_band2's whatever the run's length; a WIDE window costs a pass of its own here, while k_invert_band2 first narrows it to its live arc)
                if (ROLE == 1 && handed) {  // the second band kernel's
                    myc = NC;
                    eligb = false;
                    flags |= F_TO_B;
                }
                if (ROLE == 2 && eligb && !handed) {  // k_invert_band has dealt with it (decided, or listed)
                    myc = NC;
                    eligb = false;
                    skip = true;
                }
            }
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
            // the pixels a class would leave for a part-filled last pass move up into the next wider class when they fit into
            // ITS part-filled last pass (their slots lie right before that class's: only the boundary moves; a narrow window in
            // a wide segment merely leaves lanes idle): one pass less each time
#pragma unroll
            for (int c = 0; c + 1 < NC; ++c) {
                const int np = 64 / (2 << (c >> 1)), npn = 64 / (2 << ((c + 1) >> 1));  // pixels per pass of this class / of the next
                const int rem = ncls[c] % np;
                const int added = (ncls[c + 1] + rem + npn - 1) / npn - (ncls[c + 1] + npn - 1) / npn;
                if (rem > 0 && added == 0) { ncls[c] -= rem; ncls[c + 1] += rem; first[c + 1] -= rem; }
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {  // wave-uniform by construction: say so (they live in SGPRs through the passes, not in VGPRs)
                first[c] = __builtin_amdgcn_readfirstlane(first[c]);
                ncls[c] = __builtin_amdgcn_readfirstlane(ncls[c]);
            }
            const bool to_rec = ROLE == 1 && A.rec_b != nullptr && (flags & F_TO_B) != 0;
            if constexpr (ROLE == 2) {
                // k_invert_band2 on a pixel WITHOUT a record (list B as indices, or a pixel marked in the strip mask because its
                // record did not fit): the record stage 1 of k_invert_band would have written, built here in registers, then
                // k_invert_band2's own search (xsw_band2.hpp).  A lane this kernel is not for (strip walk: k_invert_band kept the
                // pixel) is out; a marked pixel that is not searchable (cannot happen: both kernels run the same stage 1) is
                // passed on like any undecided one.
                const bool mine = in && !(strip_walk && (skip || (flags & F_NEED_CO) == 0));
                BandRec r;
                r.s = P.s_co; r.ah = 0.5 * P.a_re; r.bh = 0.5 * P.b_eff; r.d = __double2float_ru(W.band_d);
                r.inc_tail = P.i_inc | ((has_tail ? W.w_hi - w_hi_e : 0) << 16);
                r.rows = W.w_lo | (w_hi_e << 16); r.ipn = W.ip_lo | (ncols_p << 16);
                r.idx = (unsigned)i; r.flags = flags | F_B2_HARD;  // (no run length at hand here: refine)
                band2_core<T, TO, CR>(L, A, r, mine, mine && eligb, lane, (Band2Slot *)slots, res_, strip);
                return;
            }
            BandSlot b;
            if (eligb) {
                const double ah = 0.5 * P.a_re, bh = 0.5 * P.b_eff;
                b.sn = -P.s_co * A.inv_dsig_co; b.thr_lo = thr_lo; b.thr_hi = thr_hi;
                b.ah = ah; b.bh = bh; b.m2 = ah * ah + bh * bh;
                b.inc_bin = P.i_inc | (bin << 16); b.rows = W.w_lo | (w_hi_e << 16); b.ipn = W.ip_lo | (ncols_p << 16);
                b.bin_hi = bhi < XSW_INV_BINS ? bhi : -1;
            }
            if (ROLE == 1 && A.rec_b != nullptr) {
                // the handed pixels' RECORDS (their search parameters: k_invert_band2 then neither gathers the rasters again nor
                // redoes stage 1), appended compactly, one atomicAdd per wave; a record that does not fit leaves the pixel to the
                // overflow route (wave_tail: the list's strip mask; k_invert_band2 redoes stage 1 for the marked pixels)
                const unsigned long long hm = __ballot(to_rec);
                if (hm) {
                    unsigned at0 = 0;
                    if (lane == 0) at0 = atomicAdd(A.list_b_count, (unsigned)__popcll(hm));
                    at0 = (unsigned)__builtin_amdgcn_readfirstlane((int)at0);
                    const unsigned at = at0 + __builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u));
                    if (to_rec && at < A.list_b_cap) {
                        BandRec r;
                        r.s = P.s_co; r.ah = 0.5 * P.a_re; r.bh = 0.5 * P.b_eff; r.d = __double2float_ru(W.band_d);
                        r.inc_tail = P.i_inc | ((has_tail ? W.w_hi - w_hi_e : 0) << 16);
                        r.rows = W.w_lo | (w_hi_e << 16); r.ipn = W.ip_lo | (ncols_p << 16);
                        r.idx = (unsigned)i;
                        r.flags = (flags & ~F_TO_B) | (hard ? F_B2_HARD : 0);  // (F_B2_CROWD rides in `flags`)
                        ((BandRec *)A.rec_b)[at] = r;
                        flags |= F_REC_DONE;
                    }
                }
            }
            if (eligb) {
                slots[pos] = b;
#ifdef XSW_TIMING_STAGE1_ONLY
                res_[pos] = 0;   // (timing build: pretend the pass decided, so that nothing floods the work list)
#else
                res_[pos] = -1;
#endif
            }
        }
    }
    res_[64 + lane] = pos;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- stage 2: band passes, narrowest windows first (most pixels per pass)
#ifndef XSW_TIMING_STAGE1_ONLY  // (timing / counter builds only: results are invalid without the passes)
    {
        auto run = [&](auto seg, auto kk, int c) {
            constexpr int S = decltype(seg)::value, K = decltype(kk)::value;
            for (int p = 0; p < ncls[c]; p += 64 / S)
                co_band_pass<S, K, COUNT>(L, A.inv_dsig_co, lane, slots, res_, first[c] + p, min(64 / S, ncls[c] - p), cand);
        };
        using two = std::integral_constant<int, 2>;
        using three = std::integral_constant<int, 3>;
#ifndef XSW_BAND_WIDE_K
#define XSW_BAND_WIDE_K three  // directions per lane of the widest class (chunks of 64 K directions)
#endif
        run(std::integral_constant<int, 2>{}, two{}, 0);
        run(std::integral_constant<int, 2>{}, three{}, 1);
        run(std::integral_constant<int, 4>{}, two{}, 2);
        run(std::integral_constant<int, 4>{}, three{}, 3);
        run(std::integral_constant<int, 8>{}, two{}, 4);
        run(std::integral_constant<int, 8>{}, three{}, 5);
        run(std::integral_constant<int, 16>{}, two{}, 6);
        run(std::integral_constant<int, 16>{}, three{}, 7);
        run(std::integral_constant<int, 32>{}, two{}, 8);
        run(std::integral_constant<int, 32>{}, three{}, 9);
        run(std::integral_constant<int, 64>{}, XSW_BAND_WIDE_K{}, 10);
    }
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    pos = res_[64 + lane];
    if (pos >= 0) my_flat = res_[pos];  // -1: undecided by its pass
    if (ROLE == 1 && (flags & F_REC_DONE) != 0) in = false;  // its record is k_invert_band2's: no cross-pol search, no list, no store here
    if (ROLE == 2 && strip_walk) {  // strip walk: lanes without a co-pol search, and the pixels k_invert_band kept, are not this kernel's
        skip = skip || (flags & F_NEED_CO) == 0;
        in = in && !skip;
    }
    wave_tail<T, TO, CR, COUNT>(L, A, i, in, lane, flags, my_flat, strip, cand);
}

template <typename T, typename TO, bool CR, bool COUNT, int ROLE = 0>
__global__ __launch_bounds__(64 * XSW_BAND_WG_WAVES, CR ? XSW_BAND_WAVES_CR : XSW_BAND_WAVES) void k_invert_band(DevTables L, KArgs A)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ BandSlot slots[XSW_BAND_WG_WAVES][64];  // search parameters of the wave's eligible pixels, sorted by window class
    __shared__ int res_[XSW_BAND_WG_WAVES][128];       // [0, 64): slot -> winning flat index (or -1); [64, 128): lane -> its pixel's slot
                                                       // (parked here through the passes: one VGPR less where the pressure peaks)
    // same tile walk as k_invert (XCD x owns a contiguous range of tile columns, line groups fastest), as a 2-D grid so that
    // no division is needed: blockIdx.x = xcd + 8 * line group, blockIdx.y = tile column inside the XCD's range (workgroups
    // are dealt to the XCDs round-robin in linear order, x fastest: the XCD of a workgroup is still blockIdx.x & 7)
    const long long strips_per_line = (A.samples + 63) >> 6;
    const long long cols_per_xcd = (strips_per_line + 7) >> 3;
    const long long xcd = blockIdx.x & 7;
    const long long col = xcd * cols_per_xcd + blockIdx.y;
    const long long line = (long long)(blockIdx.x >> 3) * XSW_BAND_WG_WAVES + wv;
    if (col >= strips_per_line || line >= A.lines) return;  // wave-uniform
    const long long smp = col * 64 + lane;
    const bool in = smp < A.samples;
    const long long i = line * A.samples + (in ? smp : A.samples - 1);
    band_wave<T, TO, CR, COUNT, ROLE>(L, A, i, in, lane, slots[wv], res_[wv], false, line * strips_per_line + col);
}

}  // namespace xsw
