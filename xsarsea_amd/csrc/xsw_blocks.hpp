// `k_invert_blocks` (round 4): the block pyramid of co_block_search (xsw_device.hpp; tests/prune_model.py: block_pruned_argmin)
// with FOUR pixels per wave at a time, one per 16-lane segment -- the kernel for the pixels the band rule does not pay for:
// windows that leave the monotone rows of the LUT, bands that hold thousands of candidates (an a-priori wind far from the sigma0
// contour), sigma0 outliers whose window is the whole grid.  k_invert_band appends them to list C.
//
// Why segments: one such search is a chain of dependent round trips (band bounds -> block bounds -> candidates) with little
// arithmetic in between; a wave that walks its pixels one after the other (k_invert_list) waits for memory most of the time and
// pays the fixed cost of every step -- ballots, reductions, scalar control -- per pixel.  With a pixel per segment the steps of
// four searches share that cost and keep four times the loads in flight; a block of XSW_BLK_R = 4 speed rows x XSW_BLK_C = 16
// directions is exactly one segment wide (lane = direction, four rows per lane: four more loads in flight).
//
// One pass (co_seg16_pass), per segment:
//   level 1   (windows of more than XSW_SEG_DIRECT blocks)  lower bound of every band of L.blk_g block rows the window touches;
//   round 1   the blocks of the most promising band (or of the whole window, when it is small) are bounded, the kept ones queued
//             in LDS and swept: the running minimum tightens the bound;
//   round 2   the bands that survive the tightened bound: their blocks bounded, queued, swept;
//   settle    a unique candidate within eps of the segment's screening minimum is the reference's argmin; anything else
//             (near-ties, a queue that overflowed, more than 64 bands) leaves the pixel to k_invert_list, whose wave-wide block
//             search re-scores near-ties in the reference's operation order.
// The bounds, their deflation and the exactness argument are co_block_search's.
#pragma once
#include "xsw_band2.hpp"

namespace xsw {

#ifndef XSW_SEGQ_CAP
#define XSW_SEGQ_CAP 64  // kept blocks a segment queues before they are swept (and the bound tightened)
#endif
#ifndef XSW_FINEQ_CAP
#define XSW_FINEQ_CAP 64  // surviving quarter blocks a segment collects before they are swept
#endif
#ifndef XSW_SEG_DIRECT
#define XSW_SEG_DIRECT 96  // windows of at most this many blocks are bounded block by block, without the band level
#endif
#ifndef XSW_BLOCKS_WAVES
#define XSW_BLOCKS_WAVES 4
#endif
static_assert(XSW_BLK_R == 4 && XSW_BLK_C == 16, "a block is one 16-lane segment wide, four rows per lane");

struct SegQEntry { int brbc; float lb; };  // block row | block column << 16; its lower bound, rounded down

__device__ __forceinline__ unsigned seg_bits(unsigned long long m, int q) { return (unsigned)(m >> (q * 16)) & 0xffffu; }

// lower bound of J over block (br, bc) of a slice: the sigma0 term from the block's {min, max}, the wind term from the distance
// of m/2 to the block's polar cell (co_block_search; tests/prune_model.py: sig_lb + cell_wind_lb).
// FLOAT32, deflated (round 5): it is a bound, and the float64 form with libm's fmax / fmin cost 165 VALU instructions per step of
// 16 blocks where the sweep of a block costs 250 -- a third of the kernel.  The block tables are float32 already; everything that
// goes into the bound is rounded once more (~1e-7 of its magnitude), the result is lowered by XSW_BOUND_SLACK x those magnitudes,
// and a direction within 1e-5 |m| of a cell's edge counts as inside it (the smaller bound).
struct BlockBound32 { float s, ainv, ah, bh, m2, mh, tol, wh0, whs; };
template <int C = XSW_BLK_C, int RR = XSW_BLK_R>
__device__ __forceinline__ float block_lb(const DevTables &L, float2 mm, int br, int bc, const BlockBound32 &q, bool span_ok)
{
    const int r0 = min(br * RR, L.n_w - 1), r1 = min(br * RR + RR, L.n_w) - 1;
    const int c0 = min(bc * C, L.n_phi - 1), c1 = min(bc * C + C, L.n_phi) - 1;
    const float wha = fmaf((float)r0, q.whs, q.wh0), whb = fmaf((float)max(r1, r0), q.whs, q.wh0);
    const float dsg = vmaxf(0.0f, vmaxf(mm.x - q.s, q.s - mm.y)) * q.ainv;
    const float rad = vmaxf(0.0f, vmaxf(wha - q.mh, q.mh - whb));
    float lbw = rad * rad;
    if (span_ok) {
        const float2 ea = ((const float2 *)L.csphi32)[c0], eb = ((const float2 *)L.csphi32)[max(c1, c0)];
        const bool inside = (ea.x * q.bh - ea.y * q.ah >= -q.tol) && (q.ah * eb.y - q.bh * eb.x >= -q.tol);
        const float pmx = vmaxf(q.ah * ea.x + q.bh * ea.y, q.ah * eb.x + q.bh * eb.y);
        const float tt = vminf(vmaxf(pmx, wha), whb);
        const float e2 = fmaf(tt, tt - 2.0f * pmx, q.m2);
        lbw = inside ? lbw : vmaxf(lbw, e2);
    }
    const float lb = fmaf(dsg, dsg, lbw);
    return lb - XSW_BOUND_SLACK * (lb + q.m2 + whb * whb) - 1e-6f;
}

// Up to four pending pixels of the wave (lane l owns pixel l: P_*), one per segment.  Decided pixels: my_flat of the owner lane;
// the others are flagged in `redo`.
__device__ __forceinline__ void co_seg16_pass(const DevTables &L, double inv_dsig, int lane, double P_s, double P_a, double P_b, double P_bd, int P_iinc,
                                              int P_rows, int P_dirs, unsigned long long &pend, SegQEntry *__restrict__ qlds /* this wave's [4][XSW_SEGQ_CAP], then [4][XSW_FINEQ_CAP] */,
                                              int &my_flat, unsigned long long &redo, unsigned &cand)
{
    constexpr int R = XSW_BLK_R, C = XSW_BLK_C, CAP = XSW_SEGQ_CAP;
    const double inf = __builtin_inf();
    const int q = lane >> 4, sl = lane & 15;
    int o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o[k] = pend ? (__ffsll((long long)pend) - 1) : -1;
        if (pend) pend &= pend - 1;
    }
    int own = o[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) own = (q == k) ? o[k] : own;
    const bool valid = own >= 0;
    const int src = valid ? own : 0;
    const double s = __shfl(P_s, src), a = __shfl(P_a, src), b = __shfl(P_b, src), bd = __shfl(P_bd, src);
    const int i_inc = __shfl(P_iinc, src);
    const unsigned rows = (unsigned)__shfl(P_rows, src), dirs = (unsigned)__shfl(P_dirs, src);
    const double ah = 0.5 * a, bh = 0.5 * b, m2 = ah * ah + bh * bh, mh = sqrt(m2), sn = -s * inv_dsig, ainv = fabs(inv_dsig);
    const double wh0 = 0.5 * L.w0, whs = L.wstep_half;
    const double slack = 1e-8 * (1.0 + m2), tol = 1e-9 * mh + 1e-300;
    BlockBound32 Q;
    Q.s = (float)s; Q.ainv = (float)ainv * (1.0f - 1e-6f); Q.ah = (float)ah; Q.bh = (float)bh; Q.m2 = (float)m2; Q.mh = (float)mh;
    Q.tol = 1e-5f * Q.mh + 1e-30f; Q.wh0 = (float)wh0; Q.whs = (float)whs;
    const double rs = bd * ainv;  // >= sqrt(J_ub) (co_window_lanes: band_d = |dsig| sqrt(J_ub), inflated)
    double jub = rs * rs * (1.0 + 1e-12);
    const int w_lo = (int)(rows & 0xffffu), w_hi = min(max((int)(rows >> 16), w_lo), L.n_w - 1);
    const int ip_lo = (int)(dirs & 0xffffu), ip_hi = min(max((int)(dirs >> 16), ip_lo), L.n_phi - 1);
    const int br_lo = w_lo / R, br_hi = w_hi / R, bc_lo = ip_lo / C, bc_hi = ip_hi / C, ncb = bc_hi - bc_lo + 1;
    const int nbw = (br_hi - br_lo + 1) * ncb;
    // level-1 cells of XSW_CELL_R x XSW_CELL_C blocks the window touches: ncr x ncc of them, cell t = (t / ncc, t % ncc) from (cr_lo, cc_lo)
    constexpr int GR = XSW_CELL_R, GC = XSW_CELL_C;
    const int cr_lo = br_lo / GR, cc_lo = bc_lo / GC, ncc = bc_hi / GC - cc_lo + 1, ncell = (br_hi / GR - cr_lo + 1) * ncc;
    const float inv_ncc = 1.0f / (float)ncc, inv_ncb = 1.0f / (float)ncb;
    const bool bandmode = valid && nbw > XSW_SEG_DIRECT;
    bool bad = bandmode && ncell > 128;  // (LUTs of thousands of speed rows: the wave-wide search of k_invert_list)
    const char *__restrict__ base = (const char *)L.co;
    const unsigned rowB = (unsigned)L.phi_pad * 8u, slice0 = (unsigned)(i_inc * L.n_w) * rowB;
    const float2 *__restrict__ blk = L.blk + (size_t)i_inc * L.nbr * L.nbc;
    const float2 *__restrict__ cel = L.cellmm + (size_t)i_inc * L.ncr * L.ncc;
    SegQEntry *__restrict__ qseg = qlds + q * CAP;
    SegQEntry *__restrict__ fseg = qlds + 4 * CAP + q * XSW_FINEQ_CAP;  // the surviving quarter blocks of the current groups of queue entries

    double best = inf, second = inf;
    int bflat = 0;
    int qn = 0;

    // level 1, one lane per cell (cell 16 j + sl of the window's): the segment's most promising cell (mask = false), or the set of
    // cells the current bound keeps (mask = true), as a bit mask of 128
    const int nj = (wave_max_i(bandmode && !bad ? ncell : 0) + 15) >> 4;  // wave-uniform, <= 8
    auto level1 = [&](bool mask, int &first, unsigned long long &m0, unsigned long long &m1) {
        float mn = 3e38f;
        int tmin = -1;
        m0 = 0ULL;
        m1 = 0ULL;
#pragma unroll 1
        for (int j = 0; j < nj; ++j) {  // wave-uniform trip count (ONE copy of the bound's code: the kernel's size is felt in the instruction cache)
            const int t = 16 * j + sl;
            const bool tv = bandmode && !bad && t < ncell;
            const int cr = tv ? (int)(((float)t + 0.5f) * inv_ncc) : 0, cc = tv ? t - cr * ncc : 0;
            const int cra = min(cr_lo + cr, L.ncr - 1), cca = min(cc_lo + cc, L.ncc - 1);
            const float2 mm = cel[cra * L.ncc + cca];
            const float lb = block_lb<GC * C, GR * R>(L, mm, cra, cca, Q, L.cell_span_ok != 0);
            if (!mask) {  // (uniform)
                tmin = (tv && lb < mn) ? t : tmin;
                mn = (tv && lb < mn) ? lb : mn;
            } else {
                const unsigned long long bits = (unsigned long long)seg_bits(ballot64(tv && !((double)lb * (1.0 - 1e-8) > jub + slack)), q);
                if (j < 4) m0 |= bits << (16 * j);
                else m1 |= bits << (16 * (j - 4));
            }
        }
        if (!mask) {  // the segment's smallest bound: the cell of the lowest lane that holds it
            const float g = (float)seg_min_d<16>((double)mn);
            const unsigned bits = seg_bits(ballot64(tmin >= 0 && mn == g), q);
            const int t_of = __shfl(tmin, bits ? q * 16 + (__ffs((int)bits) - 1) : lane);
            first = bits ? t_of : -1;
        }
    };
    // the blocks of cell t inside the window: block rows from curA, block columns from curC, curW columns wide, curN blocks
    int curA = br_lo, curC = bc_lo, curW = ncb, curN = 0, k0 = 0;
    float inv_curW = inv_ncb;
    auto cell_range = [&](int t) {
        const int cr = (int)(((float)t + 0.5f) * inv_ncc), cc = t - cr * ncc;
        const int a0 = max((cr_lo + cr) * GR, br_lo), a1 = min(min((cr_lo + cr + 1) * GR, L.nbr) - 1, br_hi);
        const int b0 = max((cc_lo + cc) * GC, bc_lo), b1 = min(min((cc_lo + cc + 1) * GC, L.nbc) - 1, bc_hi);
        curA = a0; curC = b0; curW = max(b1 - b0 + 1, 1);
        inv_curW = curW == 1 ? 1.0f : (curW == 2 ? 0.5f : 1.0f / (float)curW);
        curN = max(a1 - a0 + 1, 0) * curW;
    };
    // bounds the blocks of the segment's current band [curA.., curN blocks, from block k0 on), then of the bands left in `bands`, 16
    // per step; the kept ones are queued.  A segment whose queue is nearly full pauses (its position is kept): the queues are
    // swept, the bound tightens, and bound_blocks is called again.
    unsigned long long bands = 0ULL, bands1 = 0ULL;
    auto bound_blocks = [&]() {
        while (ballot64(k0 < curN && qn <= CAP - 16) != 0ULL) {
            const bool go = k0 < curN && qn <= CAP - 16;
            const int idx = k0 + sl;
            const bool bv = go && idx < curN;
            const int dr = bv ? (int)(((float)idx + 0.5f) * inv_curW) : 0, dc = bv ? idx - dr * curW : 0;
            const int br = curA + dr, bc = curC + dc;
            const float2 mm = blk[bv ? br * L.nbc + bc : 0];
            const float lb = block_lb(L, mm, br, bc, Q, L.blk_span_ok != 0);
            const bool keep = bv && !((double)lb > jub + slack);
            const unsigned kb = seg_bits(ballot64(keep), q);
            if (keep) {
                SegQEntry e;
                e.brbc = br | (bc << 16);
                e.lb = lb;
                qseg[qn + __popc(kb & ((1u << sl) - 1u))] = e;
            }
            qn += __popc(kb);
            k0 += go ? 16 : 0;
            while (go && k0 >= curN && (bands | bands1) != 0ULL) {  // the segment's next cell
                const int t = bands ? __ffsll((long long)bands) - 1 : 64 + __ffsll((long long)bands1) - 1;
                if (bands) bands &= bands - 1;
                else bands1 &= bands1 - 1;
                cell_range(t);
                k0 = 0;
            }
        }
    };
    // sweeps the queued blocks: lane = direction, four speed rows per lane; the running minimum tightens the bound
    // one sweep step: lane = direction `dir` (okd: a candidate's), rows row0 .. row0 + 3
    auto sweep_rows = [&](bool act, int row0_in, int dir) {
        const int dirc = act ? min(dir, L.n_phi - 1) : 0;
        const bool okd = act && dir < L.n_phi;
        const int row0 = act ? row0_in : 0;
        const unsigned off0 = slice0 + (unsigned)dirc * 8u;
        double v[R];
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = ld_co(base, off0, min(row0 + k, L.n_w - 1), rowB);
        const double2 cs = ((const double2 *)L.csphi)[dirc];
        const double U = 2.0 * (ah * cs.x + bh * cs.y);
        int flat = row0 * L.n_phi + dirc;
        double wh = fma((double)row0, whs, wh0);
        const double snl = okd ? sn : 1e150;  // a lane without a candidate scores ~1e300: never the minimum, never within eps of it
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double dd = fma(v[k], inv_dsig, (row0 + k < L.n_w) ? snl : 1e150);
            const double J = fma(dd, dd, wh * (wh - U));
            second = vmin(second, vmax(J, best));
            bflat = J < best ? flat : bflat;
            best = vmin(best, J);
            wh += whs;
            flat += L.n_phi;
        }
    };
    auto tighten = [&]() {
        const double g = seg_min_d<16>(best);
        jub = (g < 1e290) ? fmin(jub, (g + m2) * (1.0 + 1e-9) + 1e-9) : jub;
    };
    // SUB-BLOCK sweep (round 5): where the GMF saturates sigma0 varies faster with the direction than with the speed, so a block
    // 16 directions wide nearly always straddles the contour -- its sigma0 bound is zero, and with the EXACT bound the pyramid
    // still swept 2 100 candidates per pixel (33 blocks: a-priori x 2.5; scratch study: 440 in as many quarter blocks).  A kept
    // block is bounded once more per QUARTER (XSW_BLK_C4 = 4 directions, table L.blk4: lane = quarter sl & 3 of queue entry
    // sl >> 2: four entries per step), the quarters that survive are noted (<= 16 per step) and swept four at a time (lane =
    // direction sl & 3 of quarter sl >> 2, four rows per lane as before).
    auto sweep_queue4 = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int jmax = wave_max_i(qn);
        const float2 *__restrict__ blk4 = L.blk4 + (size_t)i_inc * L.nbr * L.nbc4;
        const int sub = sl & 3, ent = sl >> 2;
        // the surviving quarters of several groups of four entries are collected (XSW_FINEQ_CAP per segment) before they are swept: a
        // sweep step takes four quarters per segment, and swept group by group the last step of every group was part-filled (5.8
        // steps per pixel at 47 % of the lanes on a-priori x 2.5)
        int fn = 0;
#pragma unroll 1
        for (int j0 = 0; j0 < jmax; j0 += 4) {
            const SegQEntry e = qseg[min(j0 + ent, CAP - 1)];
            const bool actc = j0 + ent < qn && !((double)e.lb * (1.0 - 1e-8) > jub + slack);
            const int br = e.brbc & 0xffff, bc4 = (int)((unsigned)e.brbc >> 16) * 4 + sub;
            const bool v4 = actc && bc4 * XSW_BLK_C4 < L.n_phi;
            const float2 mm4 = blk4[v4 ? br * L.nbc4 + bc4 : 0];
            const float lb4 = block_lb<XSW_BLK_C4>(L, mm4, v4 ? br : 0, v4 ? bc4 : 0, Q, L.blk_span_ok != 0);
            const bool keep4 = v4 && !((double)lb4 > jub + slack);
            const unsigned kb = seg_bits(ballot64(keep4), q);
            if (keep4) {
                SegQEntry f;
                f.brbc = br | (bc4 << 16);
                f.lb = lb4;
                fseg[fn + __popc(kb & ((1u << sl) - 1u))] = f;
            }
            fn += __popc(kb);
            const bool last = j0 + 4 >= jmax;
            if (!last && ballot64(fn > XSW_FINEQ_CAP - 16) == 0ULL) continue;  // room for another group in every segment
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int steps = (wave_max_i(fn) + 3) >> 2;
#pragma unroll 1
            for (int t = 0; t < steps; ++t) {
                const int fi = t * 4 + ent;
                const SegQEntry f = fseg[min(fi, XSW_FINEQ_CAP - 1)];
                const bool act = fi < fn && !((double)f.lb * (1.0 - 1e-8) > jub + slack);
                const unsigned long long am = ballot64(act);
                if (am == 0ULL) continue;
                sweep_rows(act, (f.brbc & 0xffff) * R, (int)((unsigned)f.brbc >> 16) * XSW_BLK_C4 + sub);
                tighten();  // (every step: tightening every other step swept 2 % more candidates and cost 3 % of the kernel)
                cand += (unsigned)__popcll(am) * 4u;
            }
            fn = 0;
            __builtin_amdgcn_wave_barrier();  // the quarter list is rewritten by the next groups
        }
        qn = 0;
        __builtin_amdgcn_wave_barrier();  // the queue is rewritten by the next round
    };
    auto sweep_queue = [&]() {
        if (L.blk4) { sweep_queue4(); return; }  // (uniform)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int jmax = wave_max_i(qn);
#pragma unroll 1
        for (int j = 0; j < jmax; ++j) {
            const SegQEntry e = qseg[min(j, CAP - 1)];
            const bool act = j < qn && !((double)e.lb * (1.0 - 1e-8) > jub + slack);
            const unsigned long long am = ballot64(act);
            if (am == 0ULL) continue;
            sweep_rows(act, (e.brbc & 0xffff) * R, (int)((unsigned)e.brbc >> 16) * C + sl);
            if ((j & 1) != 0 || j + 1 >= jmax) tighten();  // every other block: the bound follows the segment's running minimum
            cand += (unsigned)__popcll(am) * 4u;
        }
        qn = 0;
        __builtin_amdgcn_wave_barrier();  // the queue is rewritten by the next round
    };

    // round 1: the most promising band (large windows), or the whole window
    int first = -1;
    curN = (valid && !bandmode) ? nbw : 0;
    if (nj > 0) {
        level1(false, first, bands, bands1);
        bands = 0ULL;
        bands1 = 0ULL;
        if (bandmode && !bad && first >= 0) cell_range(first);
    }
    do {
        bound_blocks();
        sweep_queue();
    } while (ballot64(k0 < curN) != 0ULL);
    if (nj > 0) {  // round 2: the cells the tightened bound keeps
        int dummy;
        level1(true, dummy, bands, bands1);
        if (first >= 0 && first < 64) bands &= ~(1ULL << first);
        if (first >= 64) bands1 &= ~(1ULL << (first - 64));
        curN = 0;
        k0 = 0;
        if ((bands | bands1) != 0ULL) {
            const int t = bands ? __ffsll((long long)bands) - 1 : 64 + __ffsll((long long)bands1) - 1;
            if (bands) bands &= bands - 1;
            else bands1 &= bands1 - 1;
            cell_range(t);
        }
        do {
            bound_blocks();
            sweep_queue();
        } while (ballot64(k0 < curN) != 0ULL);
    }

    // settle, per segment
    const double gmin = seg_min_d<16>(best);
    const double T = gmin + 1e-9 * (1.0 + fabs(gmin) + m2);
    const unsigned long long amb = ballot64(valid && (second <= T || bad || !(gmin < 1e290))), surv = ballot64(valid && best <= T);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (o[k] < 0) continue;
        const unsigned long long bits = 0xffffULL << (16 * k), sv = surv & bits;
        if ((amb & bits) != 0ULL || __popcll(sv) != 1) {
            redo |= 1ULL << o[k];
        } else {
            const int flat = rd_lane_i(bflat, __ffsll((long long)sv) - 1);
            if (lane == o[k]) my_flat = flat;
        }
    }
}

// One wave's (up to) 64 listed pixels: stage 1 as in the band kernels (classification, incidence bin, three-ray bound, window),
// the segment passes, the cross-pol phase and the store (wave_tail: what is still undecided goes to k_invert_list's work list).
template <typename T, typename TO, bool CR>
__device__ __forceinline__ void blocks_wave(const DevTables &L, const KArgs &A, long long i, bool in, int lane, SegQEntry *__restrict__ qlds)
{
    int flags, my_flat = -1;
    unsigned cand = 0;
    {
        Pixel P;
        load_pixel<T, false>(L, A, i, in, P);
        flags = P.flags;
        if (CR && A.s_cr) {
            const T x = ((const T *)A.s_cr)[i];
            const T dr = A.dsig_cr ? ((const T *)A.dsig_cr)[i] : (T)0;
            if (x != x || dr != dr) flags |= F_CR_RAW_NAN;
        }
        const bool fin = in && (P.flags & F_NEED_CO) != 0 && (P.flags & F_CO_FINITE) != 0;
        unsigned long long pend = __ballot(fin), redo = 0ULL;
        if (pend) {
            bool loose = false;
            CoWindow W = co_window_lanes<XSW_BAND_RAYS, XSW_BAND_RAY_D, XSW_BAND_SEEDED != 0>(L, P, A.inv_dsig_co, fabs(A.dsig_co), loose);
            if (L.inv_rows && L.mono_rows) {
                // CONTOUR BOUND (round 5, xsw_band2.hpp): the rays look where the a-priori wind points; the best of the candidates where the
                // window's directions cross the observed sigma0 (inverse-row table: the monotone rows) is a bound nearer to the minimum --
                // a smaller disc, a tighter start for the pyramid
                const bool scan = fin && W.band_d < 1e300;
                const int mono1 = L.mono_rows[scan ? P.i_inc : 0];
                const double ah = 0.5 * P.a_re, bh = 0.5 * P.b_eff, m2 = ah * ah + bh * bh;
                const double jc = contour_scan(L, scan, P.i_inc, P.s_co, ah, bh, A.inv_dsig_co, W.ip_lo, W.ip_hi - W.ip_lo + 1, W.w_lo, min(W.w_hi, mono1 - 1));
                const double rs = W.band_d * fabs(A.inv_dsig_co), jub2 = (jc + m2) * (1.0 + 1e-9) + 1e-9;
                if (scan && jc < 1e300 && jub2 < rs * rs) {
                    const CoWindow W2 = box_from_jub(L, P.mag, P.theta, jub2);
                    W.w_lo = max(W.w_lo, W2.w_lo); W.w_hi = min(W.w_hi, W2.w_hi);
                    W.ip_lo = max(W.ip_lo, W2.ip_lo); W.ip_hi = min(W.ip_hi, W2.ip_hi);
                    W.band_d = (double)__builtin_sqrtf((float)jub2) * (1.0 + 1e-6) * fabs(A.dsig_co) + 1e-9;
                }
            }
            const int rows = W.w_lo | (W.w_hi << 16), dirs = (int)((unsigned)W.ip_lo | ((unsigned)W.ip_hi << 16));
            // (a bound so loose that its float32 square root overflowed: band_d = inf -- the wave-wide search handles it)
            if (!(W.band_d < 1e300)) pend &= ~__ballot(fin && !(W.band_d < 1e300));
            // the four pixels of a pass wait for the slowest of them: pixels of similar cost share a pass -- they are taken in the
            // order of their window size (blocks), by classes of powers of two
            const int nbw_p = ((W.w_hi >> 2) - (W.w_lo >> 2) + 1) * ((W.ip_hi >> 4) - (W.ip_lo >> 4) + 1);
            const int cls = 31 - __clz(max(nbw_p, 1));
#pragma unroll 1
            for (int c = 15; c >= 0 && pend; --c) {
                unsigned long long m = pend & __ballot(c == 15 ? cls >= 15 : cls == c);
                if (c > 0 && __popcll(m) < 4 && (pend & ~m)) {  // a part-filled pass: filled up from the next class
                    const unsigned long long nxt = pend & __ballot(cls == c - 1);
                    unsigned long long take = nxt;
                    for (int k = __popcll(m); k < 4 && take; ++k) { m |= take & (0ULL - take); take &= take - 1; }
                }
                pend &= ~m;
                while (m) co_seg16_pass(L, A.inv_dsig_co, lane, P.s_co, P.a_re, P.b_eff, W.band_d, P.i_inc, rows, dirs, m, qlds, my_flat, redo, cand);
            }
        }
    }
    wave_tail<T, TO, CR, true>(L, A, i, in, lane, flags, my_flat, -1, cand, 5);  // (COUNT: the statistics are a run-time switch here)
}

// Third kernel of the chain: the pixels k_invert_band left on list C, 64 per wave (fewer when the list is short: the passes of
// a wave take its pixels four at a time), fixed grid, every wave strides over the list.
template <typename T, typename TO, bool CR>
__global__ __launch_bounds__(256, XSW_BLOCKS_WAVES) void k_invert_blocks(DevTables L, KArgs A)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ SegQEntry qlds[4][4 * XSW_SEGQ_CAP + 4 * XSW_FINEQ_CAP];
    const long long count = (long long)*A.list_c_count;
    const long long nlist = count < (long long)A.list_c_cap ? count : (long long)A.list_c_cap;  // (what did not fit went to k_invert_list's list)
    const long long nwaves = (long long)gridDim.x * 4;
    int ppw = 64;
    if (nlist < 64LL * nwaves) {
        ppw = 4;
        while (ppw < 64 && (long long)ppw * nwaves < nlist) ppw <<= 1;
    }
    for (long long c = (long long)blockIdx.x * 4 + wv; c * ppw < nlist; c += nwaves) {  // wave-uniform
        const long long k = c * ppw + lane;
        const bool in = lane < ppw && k < nlist;
        const long long i = (long long)A.list_c[in ? k : nlist - 1];
        blocks_wave<T, TO, CR>(L, A, i, in, lane, qlds[wv]);
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace xsw
