// Device code of libxsw: per-pixel GMF wind inversion for gfx950 (wave64).
//
// Work decomposition:
//   * a workgroup (4 waves) owns a raster tile of 4 lines x 64 samples; one wavefront owns a strip of 64
//     consecutive samples; lane i loads pixel i (coalesced), converts sigma0 to dB, finds its incidence bin;
//   * PER-LANE stages (64 pixels in flight, wave-uniform trip counts, nothing diverges): upper bound of the
//     cost along the a-priori direction (bisection) and the search window + lane layout it implies
//     (co_window_lanes, box_from_jub, chunk_geom), the whole cross-pol search (search_cr_interval /
//     search_cr_lanes), forming and storing the complex winds (store_pixel);
//   * COOPERATIVE stage: windows of <= 16 directions go four pixels at a time, one per 16-lane segment
//     (co_seg_pass); the others one pixel at a time with the pixel's window and parameters wave-uniform
//     (co_box_search: readlane -> SGPRs); the lanes sweep the window's candidates and a DPP argmin picks the
//     winner; lane i keeps the winner of pixel i.
//
// All decisions are taken in float64.  The reference's argmin (windspeed/windspeed.py:220-232) is
// reproduced exactly: candidates are screened with a cheap fused form, every candidate within a
// conservative eps of the screening minimum is re-scored in the reference's operation order
// (`exact_J_co`), and ties go to the lowest flat index.  This file is compiled with
// -ffp-contract=off; fused multiply-adds appear only where written explicitly (`fma`).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace xsw {

#ifndef XSW_INV_BINS
#define XSW_INV_BINS 2048  // thresholds per slice of the inverse-row table (DevTables::inv_rows)
#endif

struct DevTables {
    // co-pol LUT, dB
    const double *co;    // [n_inc][n_w][phi_pad]   incidence-major slices (722 KB each at default size)
    const double *coT;   // [n_inc][n_phi][w_pad]   same slices transposed: one direction, all speeds
    const double *inc, *w, *wh, *phi, *cphi, *sphi;  // axes; wh = w/2; cphi/sphi = cos/sin(radians(phi))
    const double *csphi;     // [n_phi][2] the same cos / sin interleaved (one 16-byte load per direction in co_band_pass)
    const float *csphi32;    // [n_phi][2] float32 copy (bound arithmetic of k_invert_band2)
    const float *co32;       // [n_inc][n_w][phi_pad] float copy of `co` (screening of the exhaustive kernel)
    const float *wh32;       // [n_w] float(w/2)
    double co_absmax;        // max |co|: bounds the float32 screening error
    const double *out_dir;   // [2][n_phi][2]       exp(1j*deg2rad(+-phi))
    const double *abs_co;    // [n_w][n_phi]        |w*exp(1j*deg2rad(phi))|
    const double *dual_dir;  // [2][n_w][n_phi][2]  exp(1j*angle(sol / sol_2))
    const double *sol;       // [2][n_w][n_phi][2]  w * exp(1j*deg2rad(+-phi)): the co-pol winds themselves (k_expand)
    int n_inc, n_w, n_phi, phi_pad, w_pad;
    int phi_180;   // windspeed.py:152-156
    int prunable;  // uniform axes, finite LUT: branch-and-bound allowed
    int co_off32;  // the padded co table is < 4 GB: 32-bit byte offsets from its base address every word
    int band_mul24;  // n_inc * n_w, n_inc * XSW_INV_BINS + XSW_INV_BINS and the row pitches fit 24 bits: the band kernels' offsets by v_mul_u32_u24
    const double *tail_min;  // [n_inc][XSW_TAIL_LEVELS + 1][phi_pad] sparse table over the directions of the smallest LUT value in rows >= mono_rows[i] (k_tail_min); nullable
    const int *mono_rows;  // [n_inc] every column of slice i is non-decreasing in wind speed over rows [0, mono_rows[i]) (band pruning)
    // inverse of the monotone rows (band pruning): inv_rows[i][b][p] = first row r < mono_rows[i] of column p with
    // LUT >= fma(b, inv_grid[3i+1], inv_grid[3i]) (a uniform dB grid per slice, XSW_INV_BINS bins), else mono_rows[i]
    const unsigned short *inv_rows;  // [n_inc][XSW_INV_BINS][phi_pad]
    const double *inv_grid;          // [n_inc][3]: t0, bin width, 1 / bin width
    // block pyramid (co_block_search): blk[i][br][bc] = {min, max} of the LUT over speed rows [XSW_BLK_R br, +XSW_BLK_R) x directions
    // [XSW_BLK_C bc, +XSW_BLK_C) of slice i, float32 rounded outward; bandmm[i][t] the same over block rows [blk_g t, +blk_g), every
    // direction (blk_g = the block rows one wave trip covers, one lane per block).  Null: not built.
    const float2 *blk;     // [n_inc][nbr][nbc]
    const float2 *bandmm;  // [n_inc][nbands]
    const float2 *blk4;    // [n_inc][nbr][nbc4]: the same per SUB-BLOCK of XSW_BLK_R rows x XSW_BLK_C4 directions (k_invert_blocks, round 5); null: not built
    int nbc4;
    const float2 *cellmm;  // [n_inc][ncr][ncc]: the same per CELL of XSW_CELL_R block rows x XSW_CELL_C block columns (level 1 of k_invert_blocks, round 5); with blk or not at all
    int ncr, ncc, cell_span_ok;
    int nbr, nbc, blk_g, nbands;
    int blk_span_ok;       // a block spans less than 170 deg of direction: the sector bound of co_block_search holds
    double w0, inv_wstep, phi0, phi_last, inv_dphi;
    double wstep_half;  // 0.5 / inv_wstep (host: one IEEE division instead of one per wave and pass)
    double inv_nphi;    // 1 / n_phi: flat index -> (row, direction) without an integer division
    int inc_uniform;    // incidence axis uniform (xsw.hip uniform_axis): nearest_index starts from the computed bin
    double inc0, inv_incstep;
    // cross-pol LUT, dB
    const double *cr;    // [n_inc_cr][wcr_pad]
    const double *inc_cr, *wcr, *wcrh;
    int n_inc_cr, n_wcr, wcr_pad, cr_finite;
    int cr_monotone;   // every row non-decreasing in wind speed and the speed axis uniform: interval pruning allowed
    double wcr0, inv_wcrstep, wcrstep_half;
    // inverse of the (monotone) cross-pol rows: inv_cr[i][b] = first k with cr[i][k] >= fma(b, grid[3i+1], grid[3i]), else n_wcr
    const unsigned short *inv_cr;  // [n_inc_cr][XSW_INV_BINS]   (null: not monotone / too long: search_cr_interval is used)
    const double *inv_cr_grid;     // [n_inc_cr][3]: t0, bin width, 1 / bin width
    int inc_cr_uniform;
    double inc_cr0, inv_inccrstep;
};

#ifndef XSW_TAIL_LEVELS
#define XSW_TAIL_LEVELS 7  // levels of the sparse table of tail minima (DevTables::tail_min): windows of up to 2^7 - 1 directions
#endif
struct KArgs {
    const void *inc, *s_co, *s_cr, *dsig_cr, *anc;
    void *out_co, *out_cr;
    int *out_idx;
    unsigned *code_co, *code_cr;  // nullable: the answer as 4-byte grid codes (xsw.h: xsw_invert_args.out_code_*)
    unsigned long long *stats;  // [8]: pixels_co, cand_co, pixels_exact, pixels_cr; chain mode: [4] candidates scored by k_invert_band2, [5] k_invert_blocks, [6] k_invert_list, [7] records k_invert_band2 refined (nullable)
    int stats_chain;            // the statistics are those of the production chain (xsw_stats_enable(ctx, 2)): k_invert_band hands over as usual
    unsigned *list;             // two-kernel path (nullable): k_invert_band appends the flat index of every pixel it leaves
    unsigned *list_count;       // undecided; k_invert_list then inverts exactly those, 64 per wave
    unsigned list_cap;          // entries the list holds; the counter runs on past it (overflow: k_invert_list takes every tile)
    unsigned *list_b, *list_b_count;  // list B (nullable): pixels k_invert_band hands to k_invert_band2 (long runs of band rows; rise-then-fall columns)
    unsigned list_b_cap;
    void *rec_b;                // nullable: list B as RECORDS (BandRec, xsw_band.hpp: the pixel's search parameters as stage 1 of k_invert_band found
                                // them, 48 bytes) instead of pixel indices -- k_invert_band2 then neither gathers the rasters again nor redoes stage 1
    unsigned *list_c, *list_c_count;  // list C (nullable): finite pixels the band rule is not for, k_invert_band -> k_invert_blocks (block pyramid, four pixels per wave at a time)
    unsigned list_c_cap;
    // one 64-bit word per strip of 64 samples (strip = line * ceil(samples / 64) + strip column; nullable): bit l of mask_g =
    // pixel l of the strip is left to k_invert_list, of mask_b = handed to k_invert_band2, and DID NOT FIT into the list.  Zeroed
    // before every launch; a producer ORs a pixel in only when its append falls past the list's capacity, the consumer takes
    // the list and then -- if the counter ran past the capacity -- the marked pixels of the marked strips, instead of redoing
    // the whole raster.
    unsigned long long *mask_g, *mask_b;
    int long_run;               // k_invert_band, ROLE 1: rows along the a-priori direction from which a pixel is handed to k_invert_band2
    int tail_max;               // rows past the monotone ones a window may hold for k_invert_band2's tail sweep (0: never)
    int area_max;               // k_invert_band, ROLE 1: band candidates (run x directions) beyond which a pixel skips k_invert_band2 (general kernel instead)
    int b2_crowd, area_crowd_max;  // k_invert_band, ROLE 1: pixels beyond area_max stay k_invert_band2's when b2_crowd or more of the wave's 64 are such (and their area is at most area_crowd_max)
    int block_min;              // general kernel: windows of at least this many candidates are searched by the block pyramid (co_block_search)
    int arc_min, arc_crowd;     // k_invert_band: windows of at least arc_min directions are narrowed to their live arc in stage 1 when arc_crowd or more of the wave's 64 pixels are such (0x7fffffff: never)
    int wide_min;               // k_invert_band, ROLE 1: windows of at least this many directions are handed to k_invert_band2 whatever their run (its live arc narrows them)
    int b2_refine_min;          // k_invert_band2: records marked F_B2_HARD a wave of 64 must hold for the wave to run the refinement (it costs every lane of the wave)
    int b2_rows_max;            // k_invert_band2: rows the live arc of a pixel may hold after the joint shrink's first step before the pixel is passed on to k_invert_blocks
    long long n, lines, samples;
    double dsig_co, inv_dsig_co, dsig_cr_scalar;
    int is_db, dual_select;
};

// grid codes (xsw.h): bits 0..29 flat index (i_wspd * n_phi + i_phi; cross-pol: i_wspd_cr), bit 30 the -phi solution
// (cross-pol: bit 30 = the dual select picked the co-pol wind, index 0x3FFFFFFF = no cross-pol search ran)
enum : unsigned { K_CODE_NAN_RE = 0xFFFFFFFFu /* (nan, 0) */, K_CODE_NAN = 0xFFFFFFFEu /* (nan, nan) */,
                  K_CODE_PICK_CO = 0x40000000u, K_CODE_NO_INDEX = 0x3FFFFFFFu };

enum : int { F_NEED_CO = 1, F_NEED_CR = 2, F_EARLY_NAN = 4, F_CO_FINITE = 8, F_CR_RAW_NAN = 16 /* band kernel: a raw cross-pol input is NaN */,
             F_CO_LOOSE = 32 /* general kernel: finite inputs, but a bound far above the scale of the scores: block pyramid, no forward differences */,
             F_REC_DONE = 256 /* k_invert_band: the pixel's record is on list B: nothing more to do for it in this wave */,
             F_B2_HARD = 512 /* list B's record: a long run x wide window, or a tail -- worth k_invert_band2's refinement (contour bound, live arc) */,
             F_B2_CROWD = 1024 /* list B's record: beyond XSW_B2_AREA band candidates, kept for k_invert_band2 because its wave was crowded with such pixels: its wave there MUST refine */,
             F_TO_B = 64, F_TO_C = 128 /* band kernels: the pixel is list B's (k_invert_band2) / list C's (k_invert_blocks) if it is still undecided at the end of the wave */ };

// ------------------------------------------------------------------------------------------------
// wave64 helpers
__device__ __forceinline__ int rd_lane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double rd_lane_d(double v, int l)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// v_min_f64 / v_max_f64 without the NaN-canonicalisation moves clang adds around fmin/fmax
// (callers guarantee non-NaN operands).
__device__ __forceinline__ double vmin(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmax(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// uniform (SGPR) x per-lane, both below 2^24: one full-rate v_mul_u32_u24 (the compiler turns __umul24 back into the quarter-rate
// v_mul_lo_u32 when it can prove one operand short)
__device__ __forceinline__ unsigned mul24_sv(unsigned uniform, unsigned v)
{
    unsigned r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "s"(uniform), "v"(v));
    return r;
}
// DPP lane permutes (no LDS traffic): ctrl codes of the gfx9 family -- quad_perm 0x00-0xFF,
// row_half_mirror 0x141, row_mirror 0x140, row_bcast:15 0x142, row_bcast:31 0x143.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    if (ROW_MASK == 0xF)  // every lane is written (quad_perm / mirrors always have a source): no `old` operand to set up
        return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true),
                                __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true));
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xF, false),
                            __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xF, false));
}
// minimum over the 64 lanes, returned wave-uniform (every lane must be active)
__device__ __forceinline__ double wave_min_d(double v)
{
    v = vmin(v, dpp_d<0xB1, 0xF>(v));   // lane ^ 1
    v = vmin(v, dpp_d<0x4E, 0xF>(v));   // lane ^ 2
    v = vmin(v, dpp_d<0x141, 0xF>(v));  // 8-lane halves mirrored
    v = vmin(v, dpp_d<0x140, 0xF>(v));  // 16-lane rows mirrored
    v = vmin(v, dpp_d<0x142, 0xA>(v));  // lane 15 of rows 0,2 -> rows 1,3
    v = vmin(v, dpp_d<0x143, 0xC>(v));  // lane 31 -> rows 2,3
    return rd_lane_d(v, 63);
}
__device__ __forceinline__ float vminf(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmaxf(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v)
{
    const int x = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_min_f(float v)
{
    v = vminf(v, dpp_f<0xB1, 0xF>(v));
    v = vminf(v, dpp_f<0x4E, 0xF>(v));
    v = vminf(v, dpp_f<0x141, 0xF>(v));
    v = vminf(v, dpp_f<0x140, 0xF>(v));
    v = vminf(v, dpp_f<0x142, 0xA>(v));
    v = vminf(v, dpp_f<0x143, 0xC>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// lexicographic (J, idx) minimum over the wave; J never NaN here
__device__ __forceinline__ void wave_argmin(double &J, int &idx)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double oJ = __shfl_xor(J, off);
        int oI = __shfl_xor(idx, off);
        bool take = (oJ < J) || (oJ == J && oI < idx);
        J = take ? oJ : J;
        idx = take ? oI : idx;
    }
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_min_i(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
    return v;
}

// ------------------------------------------------------------------------------------------------
// sigma0 -> dB exactly as invert_from_model does on the host (windspeed.py:126-130): the arithmetic
// runs in the raster's dtype; log10 itself is evaluated in float64 and rounded once.
__device__ __forceinline__ double log10_fast(double v);
__device__ __forceinline__ double to_db(float x, int is_db)
{
    if (is_db) return (double)x;
    float y = x + 1e-15f;
    // float32 log10 rounded from a float64 one good to ~1e-15: the correctly rounded value but for ~2e-8 of the arguments
    float l = (float)log10_fast((double)y);
    return (double)(10.0f * l);
}
__device__ __forceinline__ double to_db(double x, int is_db)
{
    if (is_db) return x;
    return 10.0 * log10(x + 1e-15);
}

// np.argmin(np.abs(dim - x)) for a strictly ascending axis (windspeed.py:212, :254): first minimum.
// `uniform` (host: the axis is uniform to 1e-12): the lower bound is looked for from the computed bin (a step or two, two
// loads in flight) instead of by a bisection of dependent loads; the invariant  dim[lo-1] < x <= dim[lo]  is what decides either way.
__device__ __forceinline__ int nearest_index(const double *__restrict__ dim, int n, double x, bool uniform = false, double x0 = 0.0,
                                             double inv_step = 0.0)
{
    if (isinf(x)) return 0;  // every |dim - x| is inf: argmin returns 0
    int lo = 0, hi = n;
    if (uniform) {
        const double g = fmin(fmax((x - x0) * inv_step, -1.0), (double)n);  // NaN -> -1 (callers never pass one)
        int k = min(max((int)ceil(g), 0), n);
        bool ok = false;
#pragma unroll 1
        for (int it = 0; it < 4 && !ok; ++it) {
            const double below = dim[max(k - 1, 0)], here = dim[min(k, n - 1)];
            const bool down = k > 0 && !(below < x), up = k < n && here < x;
            k += (up ? 1 : 0) - (down ? 1 : 0);
            ok = !up && !down;
        }
        if (ok) lo = hi = k;  // else: not where the arithmetic said (cannot happen on a uniform axis): bisect
    }
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (dim[mid] < x) lo = mid + 1; else hi = mid;
    }
    if (lo == 0) return 0;
    if (lo == n) return n - 1;
    double dl = fabs(dim[lo - 1] - x), dh = fabs(dim[lo] - x);
    return (dh < dl) ? lo : lo - 1;
}

// log10 of a positive normal finite double to ~1e-15 relative (frexp + atanh series, reciprocal + Newton instead of the
// IEEE division): a third of the instructions of the library call; anything else goes to the library.
__device__ __forceinline__ double log10_fast(double v)
{
    if (!(v >= 2.2250738585072014e-308 && v <= 1.7976931348623157e308)) return log10(v);  // 0, negative, denormal, inf, NaN
    int e;
    double m = frexp(v, &e);  // [0.5, 1)
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }  // [sqrt(1/2), sqrt(2))
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
    // log10(v) = e log10(2) + ln(m) / ln(10), log10(2) split so that e * hi is exact (hi has 40 bits)
    return fma((double)e, 0.30102999566361177, fma((double)e, 3.694239077158931e-13, lnm * 0.4342944819032518));
}

// ------------------------------------------------------------------------------------------------
// exact scoring, reference operation order
__device__ __forceinline__ double exact_J_co(double w, double c, double s_, double lutv, double s, double a,
                                             double b, double dsig)
{
    double t1 = (w * c - a) * 0.5;   // (x)/2 == x*0.5 exactly
    double t2 = (w * s_ - b) * 0.5;
    double jw = t1 * t1 + t2 * t2;
    double d = (lutv - s) / dsig;
    return jw + d * d;
}
__device__ __forceinline__ double exact_J_cr(double wc, double lutv, double s, double dsig, bool have_co,
                                             double aco)
{
    double d = (lutv - s) / dsig;
    double J = d * d;                // Jsig_cr
    if (have_co) {
        double t = (wc - aco) * 0.5;
        J = J + t * t;               // Jsig_cr + Jwind_cr  (windspeed.py:261)
    }
    return J;
}

// Full (wspd x phi) sweep in the reference's arithmetic with numpy.argmin semantics (first minimum;
// a NaN anywhere wins, first NaN).  Wave-cooperative, every argument wave-uniform.  Any LUT.
// Lanes = directions (chunks of 64), XSW_EXACT_ROWS rows per trip with all their loads in flight before the first score (a
// scan of one dependent load per trip is a chain of ~1500 memory round trips: 1.5 ms for ONE pixel, which was the whole of
// k_invert_list's time on the benchmark scene -- a dozen such pixels, one wave each).  The order of the visits does not
// matter: the running minimum is lexicographic in (J, flat index).
#ifndef XSW_EXACT_ROWS
#define XSW_EXACT_ROWS 4
#endif
__device__ __forceinline__ int exact_scan_co(const DevTables &L, int i_inc, double s, double a, double b,
                                          double dsig, int lane)
{
    const double *__restrict__ slice = L.co + (size_t)i_inc * L.n_w * L.phi_pad;
    double bestJ = __builtin_inf();
    int bestI = 0x7fffffff, nanI = 0x7fffffff;
    for (int ip0 = 0; ip0 < L.n_phi; ip0 += 64) {  // wave-uniform
        const int ip = ip0 + lane;
        const bool valid = ip < L.n_phi;
        const int ipc = valid ? ip : 0;
        const double c = L.cphi[ipc], sn = L.sphi[ipc];
        const double *__restrict__ col = slice + ipc;
#pragma unroll 1
        for (int iw0 = 0; iw0 < L.n_w; iw0 += XSW_EXACT_ROWS) {
            double v[XSW_EXACT_ROWS];
#pragma unroll
            for (int u = 0; u < XSW_EXACT_ROWS; ++u) v[u] = col[(size_t)min(iw0 + u, L.n_w - 1) * L.phi_pad];
#pragma unroll
            for (int u = 0; u < XSW_EXACT_ROWS; ++u) {
                const int iw = iw0 + u;
                if (iw < L.n_w && valid) {  // (iw: wave-uniform)
                    const double J = exact_J_co(L.w[iw], c, sn, v[u], s, a, b, dsig);
                    const int flat = iw * L.n_phi + ip;
                    if (J != J) nanI = min(nanI, flat);
                    else if (J < bestJ || (J == bestJ && flat < bestI) || bestI == 0x7fffffff) { bestJ = J; bestI = flat; }
                }
            }
        }
    }
    nanI = wave_min_i(nanI);
    if (nanI != 0x7fffffff) return nanI;
    wave_argmin(bestJ, bestI);
    return bestI;
}

__device__ __forceinline__ int exact_scan_cr(const DevTables &L, int i_inc, double s, double dsig, bool have_co,
                                          double aco, int lane)
{
    const double *__restrict__ row = L.cr + (size_t)i_inc * L.wcr_pad;
    double bestJ = __builtin_inf();
    int bestI = 0x7fffffff, nanI = 0x7fffffff;
    for (int k = lane; k < L.n_wcr; k += 64) {
        double J = exact_J_cr(L.wcr[k], row[k], s, dsig, have_co, aco);
        if (J != J) nanI = min(nanI, k);
        else if (J < bestJ || bestI == 0x7fffffff) { bestJ = J; bestI = k; }
    }
    nanI = wave_min_i(nanI);
    if (nanI != 0x7fffffff) return nanI;
    wave_argmin(bestJ, bestI);
    return bestI;
}

// ------------------------------------------------------------------------------------------------
// Per-lane pixel state shared by the kernels.
struct Pixel {
    double s_co, s_cr, dsig, a_re, a_im, b_eff;
    double mag, theta;  // |(a_re, b_eff)| and its direction in degrees within [phi0, phi0 + 360)
    int flags, i_inc, i_inc_cr, ipr;
};

// Branch-and-bound co-pol search (tests/prune_model.py is the executable specification), in two stages.
//
// Stage 1, `co_window_lanes`: ONE PIXEL PER LANE (64 pixels at once, wave-uniform trip count).  Upper bound
// J_ub = the minimum of the score along the direction nearest to the ancillary wind, found by bisection on the
// slope of J along that LUT column (transposed slice: a column is contiguous), then the polar bounding box of the disc
// |c - m| <= 2 sqrt(J_ub) in index space (box_from_jub).
struct CoWindow {
    int w_lo, w_hi, ip_lo, ip_hi;
    int geom, mdiv;  // lane layout of the first direction chunk (chunk_geom), precomputed one pixel per lane
    double band_d;   // |LUT - s| <= band_d is necessary for the argmin (co_band_pass)
};
// Lane layout of one direction chunk of `width` (<= S) columns and `nrows` speed rows on a segment of S lanes:
// G = S / width speed rows side by side (lane = grp * width + col; lanes >= G * width idle), swept two row groups
// per trip.  geom = G | trips << 8;  mdiv = ceil(65536 / width), so that grp = (lane * mdiv) >> 16 exactly for
// lane < 64.  A window of <= 16 / <= 32 directions is laid out for a 16- / 32-lane segment (co_seg_pass), wider
// ones for the whole wave (co_box_search).
// 32-lane segments (two pixels per pass, windows of 17..32 directions) measured slower than the whole-wave layout on the
// default LUT (234 ms vs 220 ms: G drops from 3 rows side by side to 1); 16-lane segments are a clear gain on narrow
// windows (resolution="low": 136 -> 110 ms).
#ifndef XSW_MAX_FD_TRIPS
#define XSW_MAX_FD_TRIPS 256
#endif
#ifndef XSW_SEG8
#define XSW_SEG8 1
#endif
#ifndef XSW_SEG4
#define XSW_SEG4 1
#endif
__device__ __forceinline__ int seg_lanes(int width)
{
    return (XSW_SEG4 && width <= 4) ? 4 : ((XSW_SEG8 && width <= 8) ? 8 : (width <= 16 ? 16 : 64));
}
__device__ __forceinline__ void chunk_geom(int width, int nrows, int S, int &geom, int &mdiv)
{
    const int G = S / width, step = 2 * G;
    geom = G | (((nrows + step - 1) / step) << 8);
    mdiv = (65536 + width - 1) / width;
}
// Polar bounding box (index space) of the disc |c - m| <= 2 sqrt(jub) around the ancillary wind m = mag*e^{i theta}:
// every candidate whose wind term alone is <= jub lies inside (|c - m| >= | |c| - |m| | for the speeds;
// a point of the disc is at most asin(R/|m|) away from theta in direction).  Float64 throughout, so the only
// slack needed is MRG index units: the axes are uniform to 1e-6 of a step (host-checked), the trig tables
// good to 1e-12, R is inflated by 1e-9, and the arithmetic errs by ~1e-13 -- all far below 1e-5.  One pixel per
// lane: the cost of the double asin is shared by 64 pixels.
__device__ __forceinline__ CoWindow box_from_jub(const DevTables &L, double mag, double theta, double jub)
{
    CoWindow W;
    W.w_lo = 0; W.w_hi = L.n_w - 1; W.ip_lo = 0; W.ip_hi = L.n_phi - 1;
    W.geom = 0; W.mdiv = 0; W.band_d = 0.0;
    // window geometry from float32 square root / arcsine / arctangent (a few instructions each instead of the ~20 / 70 / 90 of
    // their float64 library forms: stage 1 is VALU-bound, profiles/r03_stage1_counters.json), with the margins widened to cover
    // them -- a window may only ever grow: 2e-3 index units (mag from a float32 square root: 6e-8 relative of <= 80 m/s is
    // 5e-5 index units; theta from atan2f: ~2e-5 deg), R inflated by 1e-6, the half angle by 2e-4 deg, and no arcsine above
    // R / |m| = 0.999 (all directions instead: the arcsine's slope there would magnify the ratio's rounding)
    const double MRG = 2e-3;
    const double R = 2.0 * (double)__builtin_sqrtf((float)jub) * (1.0 + 1e-6) + 1e-6;
    if (mag < 1e6 && R < 1e6) {
        const double nw = (double)L.n_w, np_ = (double)L.n_phi;
        const double xl = (mag - R - L.w0) * L.inv_wstep, xh = (mag + R - L.w0) * L.inv_wstep;
        W.w_lo = max((int)ceil(fmin(fmax(xl - MRG - 1e-9 * fabs(xl), -4.0), nw + 4.0)), 0);
        W.w_hi = min((int)floor(fmin(fmax(xh + MRG + 1e-9 * fabs(xh), -4.0), nw + 4.0)), L.n_w - 1);
        if (R < mag * 0.999) {
            const double half = (double)asinf((float)(R / mag)) * 57.29577951308232 + 2e-4;
            double yl = (theta - half - L.phi0) * L.inv_dphi, yh = (theta + half - L.phi0) * L.inv_dphi;
            yl -= MRG + 1e-9 * fabs(yl);
            yh += MRG + 1e-9 * fabs(yh);
            const int plo = (int)ceil(fmin(fmax(yl, -4.0), np_ + 4.0));
            const int phi_i = (int)floor(fmin(fmax(yh, -4.0), np_ + 4.0));
            if (L.phi_last - theta <= 179.9 && theta - L.phi0 <= 179.9) {
                W.ip_lo = max(plo, 0);
                W.ip_hi = min(phi_i, L.n_phi - 1);
            } else if (yl >= 0.0 && yh <= np_ - 1.0) {  // the whole (unrounded) window lies on the axis: no seam inside
                W.ip_lo = plo;
                W.ip_hi = phi_i;
            }
        }
    }
    {
        const int width = min(max(W.ip_hi - W.ip_lo + 1, 1), 64);
        chunk_geom(width, max(W.w_hi - W.w_lo + 1, 1), L.co_off32 ? seg_lanes(width) : 64, W.geom, W.mdiv);
    }
    return W;
}

// `loose` is set when the upper bound is so far above the scale of the pixel's own scores (J_ub > 500 (1 + |m|^2/4))
// that the rounding error the forward differences of stage 2 pick up on the far rows of the window (<= n u 2 (m2 + J_ub)
// after n <= 2 XSW_MAX_FD_TRIPS steps, u = 2^-53) could come near the 1e-9 (1 + |J_min| + m2) screening budget: such
// a pixel (sigma0 wildly at odds with the ancillary wind, or a pathological LUT) takes the exact full scan.
// NRAYS = 3: the bound is the smallest score seen on three rays, the direction nearest to the ancillary wind and the ones RAY_D
// grid directions to either side of it (any candidate bounds the minimum from above).  On the benchmark scene that takes
// the windows from 18.4 x 25.2 to 16.2 x 22.8 (directions x speeds) for two short bisections more per pixel
// (measured, band kernel at 20000^2: 1 ray 93.9 ms; 3 full rays 85.3; side rays seeded, 4 / 3 steps 84.5 / 83.9; 5 and 7 rays 87.7 / 88.8).
#ifndef XSW_STRIP_RAYS
#define XSW_STRIP_RAYS 3  // rays of the upper bound in the general kernel (invert_strip): 1 -> 3 rays takes 3-12 % off k_invert_list on the hard scenes (windows a quarter smaller for two short bisections per wave)
#endif
#ifndef XSW_RAY_SIDE_STEPS
#define XSW_RAY_SIDE_STEPS 2
#endif
// One probe of a ray: the scores of row pair `mid` (one 16-byte load).  Returns the smaller one; `right` = the second row scores
// lower than the first (the minimum of a unimodal column lies to the right of the pair's first row).
__device__ __forceinline__ double ray_probe(const double *__restrict__ ray, int mid, int n_w, double whs, double wh0, double inv_dsig,
                                            double sn, double ur, bool &right)
{
    const double2 v = *(const double2 *)(ray + 2 * mid);  // w_pad is even: the pad row is masked below
    const double wh_a = fma((double)(2 * mid), whs, wh0), wh_b = wh_a + whs;
    const double da = fma(v.x, inv_dsig, sn), db = fma(v.y, inv_dsig, sn);
    const double Ja = fma(da, da, wh_a * (wh_a - ur));
    const double Jb = (2 * mid + 1 < n_w) ? fma(db, db, wh_b * (wh_b - ur)) : __builtin_inf();
    right = Jb < Ja;
    return vmin(Ja, Jb);
}
// SEEDED (the band kernels: L.inv_rows is installed): the first ray does not bisect the whole column.  Its minimum lies next to
// the row where the column crosses the observed sigma0 -- the sigma0 term is steep (dsig_co is a fraction of a dB), the wind
// term is not -- and the inverse-row table (xsw_band.hpp) has that row: probe its row pair, then gallop away from it (strides
// 1, 2, 4, ... pairs) for as long as the slope keeps its sign, then bisect what is left.  Every probe is a real candidate, so
// a column that is not unimodal still only loosens the bound; the trip count is that of the wave's slowest lane.
template <int NRAYS = 1, int RAY_D = 2, bool SEEDED = false>
__device__ __forceinline__ CoWindow co_window_lanes(const DevTables &L, const Pixel &P, double inv_dsig, double abs_dsig, bool &loose)
{
    const double inf = __builtin_inf();
    const bool fin = (P.flags & F_CO_FINITE) != 0;
    const double a = fin ? P.a_re : 0.0, b = fin ? P.b_eff : 0.0, s = fin ? P.s_co : 0.0;
    const double mag = fin ? P.mag : 0.0, theta = fin ? P.theta : 0.0;
    const double ah = 0.5 * a, bh = 0.5 * b, m2 = ah * ah + bh * bh, sn = -s * inv_dsig;
    const double wh0 = 0.5 * L.w0, whs = L.wstep_half;
    // J along a ray is (nearly always) unimodal: a convex wind term plus the squared distance of a monotone
    // LUT column to the observed sigma0.  Bisect on the sign of its discrete slope, J(2k+1) - J(2k), over aligned
    // row pairs (one 16-byte load per step); every score seen on the way bounds the minimum from above, so a
    // column that is not unimodal merely loosens the bound.  For a unimodal column the minimum itself is seen.
    const int npairs = (L.n_w + 1) >> 1;
    double rbest = inf;
    int seed = 0;  // row pair where the first ray ended: the side rays look around it only (RAY_SIDE_STEPS bisection steps)
#pragma unroll 1
    for (int q = 0; q < NRAYS; ++q) {
        const int dq = ((q + 1) >> 1) * RAY_D + (q > 2 ? 1 : 0);  // 0, -D, +D, -(2D+1), +(2D+1), ...
        const int ipr = fin ? min(max(P.ipr + ((q & 1) ? -dq : dq), 0), L.n_phi - 1) : 0;
        const double2 csr = ((const double2 *)L.csphi)[ipr];  // one 16-byte read, not two 8-byte ones: every lane's read is a cache line of its own and the texture addresser is 80 % busy (k_invert_band: 34.0 -> 33.6 ms)
        const double ur = 2.0 * (ah * csr.x + bh * csr.y);
        // (band_mul24: the transposed table is < 4 GB and its row index < 2^24 -- a 32-bit byte offset by two full-rate multiplies
        // instead of a 64-bit multiply, three quarter-rate instructions, per ray)
        const double *__restrict__ ray =
            L.band_mul24 ? (const double *)((const char *)L.coT + mul24_sv((unsigned)L.w_pad * 8u, mul24_sv((unsigned)L.n_phi, (unsigned)(fin ? P.i_inc : 0)) + (unsigned)ipr))
                         : L.coT + ((size_t)(fin ? P.i_inc : 0) * L.n_phi + ipr) * L.w_pad;
        if (SEEDED && q == 0) {
            int lo = 0, hi = fin ? npairs : 0;
            int mid = npairs >> 1;
            if (fin) {
                const double *g = L.inv_grid + 3 * P.i_inc;
                const int bin = (int)fmin(fmax((s - g[0]) * g[2], 0.0), (double)(XSW_INV_BINS - 1));
                mid = min((int)L.inv_rows[mul24_sv((unsigned)L.phi_pad, (unsigned)(P.i_inc * XSW_INV_BINS + bin)) + (unsigned)ipr] >> 1, npairs - 1);  // (SEEDED implies band_mul24)
            }
            int stride = 1, state = 0;  // state: 0 first probe, 1 galloping right, 2 galloping left, 3 bisecting
#pragma unroll 1
            while (__ballot(lo < hi) != 0ULL) {  // wave-uniform; every trip shrinks every open bracket
                bool right;
                const double j = ray_probe(ray, mid, L.n_w, whs, wh0, inv_dsig, sn, ur, right);
                rbest = vmin(rbest, j);
                const bool open = lo < hi;
                lo = (open && right) ? mid + 1 : lo;
                hi = (open && !right) ? mid : hi;
                state = state == 0 ? (right ? 1 : 2) : ((state == 1 && right) || (state == 2 && !right)) ? state : 3;
                const int far = state == 1 ? lo + stride - 1 : hi - stride;
                mid = state == 3 ? (lo + hi) >> 1 : far;
                mid = min(max(mid, lo), max(hi - 1, 0));
                stride <<= 1;
            }
            seed = lo;
            continue;
        }
        // q > 0: any score seen is a valid upper bound, so a side ray may start from a bracket around the first ray's
        // result (the minimum moves by a row or two per degree); a minimum outside the bracket only loosens the bound
        constexpr int half = 1 << (XSW_RAY_SIDE_STEPS - 1);
        int lo = q == 0 ? 0 : max(seed - half, 0), hi = q == 0 ? npairs : min(seed + half, npairs);
        for (int it = (q == 0 ? 32 - __clz(npairs) : XSW_RAY_SIDE_STEPS); it > 0; --it) {  // wave-uniform trip count
            const int mid = min((lo + hi) >> 1, npairs - 1);
            bool right;
            const double j = ray_probe(ray, mid, L.n_w, whs, wh0, inv_dsig, sn, ur, right);
            rbest = vmin(rbest, j);
            const bool open = lo < hi;
            lo = (open && right) ? mid + 1 : lo;
            hi = (open && !right) ? mid : hi;
        }
        if (q == 0) seed = lo;
    }
    const double jub = (rbest + m2) * (1.0 + 1e-9) + 1e-9;
    loose = fin && !(jub <= 500.0 * (1.0 + m2));
    CoWindow W = box_from_jub(L, mag, theta, jub);
    // the sigma0 term alone is >= 0 as well: ((LUT - s)/dsig)^2 <= J_ub is necessary, i.e. |LUT - s| <= |dsig| sqrt(J_ub)
    // (inflated: a candidate outside scores > J_ub (1 + 1e-9), above the exact score of the ray's best candidate)
    W.band_d = (double)__builtin_sqrtf((float)jub) * (1.0 + 1e-6) * abs_dsig + 1e-9;
    return W;
}

// Stage 2, `co_box_search`: WAVE-COOPERATIVE, one pixel at a time, every argument wave-uniform (SGPRs).
// Lanes = directions (<= 64 per chunk); a narrow chunk puts G = 64 / width speed rows side by side (chunk_geom).
// Rows are taken two groups per trip, the next trip's loads in flight: the window is rounded up to a multiple
// of 2*G rows (extra rows are real candidates, scoring them is harmless) and slid down if it would leave the
// grid, so the main sweep has no per-candidate masking (slack rows after the LUT keep look-ahead loads legal).
// The screening score carries the row slot of its candidate in the low 16 mantissa bits (a relative
// perturbation <= 2^-36, far inside the 1e-9 screening budget), so the running minimum knows where it sits and
// no compare/select pair is spent on the index.  Never tag an infinity (it would become a signalling NaN):
// excluded lanes / rows score BIG, and pixels are only admitted with |s|, |a|, |b| < 1e100 (no overflow).
// Returns the flat index iw*n_phi+ip of the reference's argmin.
__device__ __forceinline__ double tag16(double J, int keep_mask /* 0xffff0000, in a VGPR */, int slot /* uniform */)
{
    int lo;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(lo) : "v"(__double2loint(J)), "v"(keep_mask), "s"(slot));
    return __hiloint2double(__double2hiint(J), lo);
}
__device__ __forceinline__ int co_box_search(const DevTables &L, int i_inc, double s, double a, double b, int w_lo,
                                             int w_hi, int ip_lo, int ip_hi, int geom, int mdiv, double dsig,
                                             double inv_dsig, int lane, unsigned &cand, bool &went_exact,
                                             bool relayout = false /* geom was made for a narrower segment */,
                                             bool *defer = nullptr /* non-null: what would take the exact full scan is reported here instead (-1 returned) */)
{
    const double inf = __builtin_inf(), BIG = 1e300;
    const double ah = 0.5 * a, bh = 0.5 * b, m2 = ah * ah + bh * bh, sn = -s * inv_dsig;
    const double wh0 = 0.5 * L.w0, whs = L.wstep_half;
    const int nrows = w_hi - w_lo + 1, ncols = ip_hi - ip_lo + 1;
    // nrows/ncols <= 0 cannot happen in exact arithmetic (stay safe).  More than XSW_MAX_FD_TRIPS trips: the forward
    // differences would accumulate too much rounding error (see co_window_lanes) -> exact full scan.
    if (nrows <= 0 || ncols <= 0 || (geom >> 8) > XSW_MAX_FD_TRIPS) {
        if (defer) { *defer = true; return -1; }
        went_exact = true;
        return exact_scan_co(L, i_inc, s, a, b, dsig, lane);
    }
    cand += (unsigned)(nrows * ncols);
    const double *__restrict__ slice = L.co + (size_t)i_inc * L.n_w * L.phi_pad;
    double best = inf, second = inf;
    int bidx = 0;  // (iw << 16) | ip
    for (int c0 = 0; c0 < ncols; c0 += 64) {
        const int width = min(64, ncols - c0);
        if (c0 > 0 || relayout) chunk_geom(width, nrows, 64, geom, mdiv);  // rare: wider than 64 directions / re-done pixel
        const int G = geom & 0xff, trips = geom >> 8;
        const int grp = (lane * mdiv) >> 16, col = lane - grp * width;
        const bool act = grp < G;
        const int ip = ip_lo + c0 + (act ? col : 0);
        const unsigned ipB = (unsigned)ip * 8u;  // uniform table base + 32-bit byte offset: no 64-bit address arithmetic
        const double U = 2.0 * (ah * *(const double *)((const char *)L.cphi + ipB) + bh * *(const double *)((const char *)L.sphi + ipB));
        const int step = 2 * G, rows_r = trips * step;
        int w_base = w_lo;
        const bool mask_rows = rows_r > L.n_w;
        if (!mask_rows && w_base + rows_r > L.n_w) w_base = L.n_w - rows_r;
        const int row0 = w_base + (act ? grp : 0);
        const double dG = (double)G * whs;
        const double wh = fma((double)row0, whs, wh0);
        double pw = wh * (wh - U), dp = dG * (2.0 * wh - U) + dG * dG;
        const double ddp = 2.0 * dG * dG;
        const double snl = act ? sn : 1e150;  // idle lanes: dd ~ 1e150, score ~ 1e300 (finite: tags stay legal)
        // wave-uniform row bases (SGPR pairs) + one 32-bit per-lane byte offset; the loads of the next trip are
        // issued before the current one is scored (the slack rows after the LUT keep the last look-ahead legal)
        const unsigned pstepB = (unsigned)(G * L.phi_pad * 8);
        const char *__restrict__ sb0 = (const char *)slice, *__restrict__ sb1 = sb0 + pstepB;
        unsigned off = (unsigned)((row0 * L.phi_pad + ip) * 8);
        const double before = best;
        int keep_mask;
        asm("v_mov_b32 %0, 0xffff0000" : "=v"(keep_mask));  // kept in a VGPR: VOP3 takes no literal
        auto sweep = [&](auto masked) {
            double v0 = *(const double *)(sb0 + off), v1 = *(const double *)(sb1 + off);
            for (int r0 = 0; r0 < rows_r; r0 += step) {
                off += 2u * pstepB;
                const double n0 = *(const double *)(sb0 + off), n1 = *(const double *)(sb1 + off);
                const double v[2] = {v0, v1};
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const double dd = fma(v[k], inv_dsig, snl);
                    double J = tag16(fma(dd, dd, pw), keep_mask, r0 + k * G);  // the score carries its row slot
                    if (masked.value) J = (row0 + r0 + k * G) < L.n_w ? J : BIG;
                    second = vmin(second, vmax(J, best));
                    best = vmin(best, J);
                    pw += dp;
                    dp += ddp;
                }
                v0 = n0;
                v1 = n1;
            }
        };
        if (mask_rows) sweep(std::true_type{}); else sweep(std::false_type{});
        if (best < before) bidx = ((row0 + (__double2loint(best) & 0xffff)) << 16) | ip;
    }

    // settle: a unique candidate within eps of the screening minimum IS the reference's argmin; several (in
    // different lanes) are re-scored in the reference's operation order; two in one lane (or nothing finite)
    // go to the exact full scan.
    const double gmin = wave_min_d(best);
    const double T = gmin + 1e-9 * (1.0 + fabs(gmin) + m2);
    if (__ballot(second <= T) != 0ULL || !(gmin < 0.5 * BIG)) {
        if (defer) { *defer = true; return -1; }
        went_exact = true;
        return exact_scan_co(L, i_inc, s, a, b, dsig, lane);
    }
    unsigned long long surv = __ballot(best <= T);
    const int first = __ffsll((long long)surv) - 1;
    const int pk0 = rd_lane_i(bidx, first);
    int eI = (pk0 >> 16) * L.n_phi + (pk0 & 0xffff);
    surv &= surv - 1;
    if (surv) {
        int iw = pk0 >> 16, ip = pk0 & 0xffff;
        double eJ = exact_J_co(L.w[iw], L.cphi[ip], L.sphi[ip], slice[iw * L.phi_pad + ip], s, a, b, dsig);
        while (surv) {
            const int l = __ffsll((long long)surv) - 1;
            surv &= surv - 1;
            const int pk = rd_lane_i(bidx, l);
            iw = pk >> 16; ip = pk & 0xffff;
            const int flat = iw * L.n_phi + ip;
            const double J = exact_J_co(L.w[iw], L.cphi[ip], L.sphi[ip], slice[iw * L.phi_pad + ip], s, a, b, dsig);
            if (J < eJ || (J == eJ && flat < eI)) { eJ = J; eI = flat; }
        }
    }
    return eI;
}

// Stage 2 from the TABLE side, `co_block_search` (round 4; tests/prune_model.py: block_pruned_argmin is its executable
// specification): WAVE-COOPERATIVE, one pixel at a time, every argument wave-uniform.  For the pixels whose a-priori wind does
// not confine the search -- sigma0 outliers (ships, land, rain cells: the disc covers the whole grid), windows of thousands of
// candidates, near-ties -- and for ANY LUT (no monotone columns needed).  The slice is cut into blocks of XSW_BLK_R speed rows x
// XSW_BLK_C directions (64 candidates: one per lane) with their min / max LUT value tabulated at install (L.blk, float32 rounded
// outward), and into bands of L.blk_g block rows over all directions (L.bandmm).  For a block,
//     LB = (max(0, lo - s, s - hi) / dsig)^2 + (distance of m/2 to the block's polar cell, half-speed units)^2
// bounds J = Jsig + Jwind from below over its candidates -- BOTH terms together, where the window and the band rule bound each
// term on its own: a sigma0 contour that runs through the disc far from m is excluded.  Level 1: one lane per band (radial
// distance only), the most promising band first; level 2: one lane per block of a kept band; level 3: a kept block is swept,
// one candidate per lane, scores formed directly (no forward differences: no limit on the window).  The bound tightens with
// the running minimum.  A block is skipped when its deflated LB exceeds the best score known, so each of its candidates scores
// strictly above an examined one in the reference's arithmetic: it can neither be the argmin nor tie with it.  Settle as
// everywhere: a unique candidate within eps of the screening minimum is the argmin; otherwise the kept blocks are swept once
// more and every candidate within eps is re-scored in the reference's operation order (lowest flat index wins ties).
#ifndef XSW_BLK_R
#define XSW_BLK_R 4
#endif
#ifndef XSW_BLK_C
#define XSW_BLK_C 16
#endif
#ifndef XSW_CELL_R
#define XSW_CELL_R 8  // block rows ...
#endif
#ifndef XSW_CELL_C
#define XSW_CELL_C 2  // ... x block columns of a level-1 cell of k_invert_blocks (32 speed rows x 32 directions: 16 blocks, one bounding step)
#endif
#define XSW_BLK_C4 4  // directions of a sub-block (k_invert_blocks: a kept block is bounded once more per quarter before it is swept)
static_assert(XSW_BLK_R * XSW_BLK_C == 64, "one candidate of a block per lane");
static_assert(XSW_BLK_C == 4 * XSW_BLK_C4, "four sub-blocks per block");
__device__ __forceinline__ int co_block_search(const DevTables &L, int i_inc, double s, double a, double b, double jub_in, int w_lo, int w_hi,
                                               int ip_lo, int ip_hi, double dsig, double inv_dsig, int lane, unsigned &cand, bool &went_exact)
{
    constexpr int R = XSW_BLK_R, C = XSW_BLK_C;
    const double inf = __builtin_inf();
    const double ah = 0.5 * a, bh = 0.5 * b, m2 = ah * ah + bh * bh, mh = sqrt(m2), sn = -s * inv_dsig, ainv = fabs(inv_dsig);
    const double wh0 = 0.5 * L.w0, whs = L.wstep_half;
    const double slack = 1e-8 * (1.0 + m2), tol = 1e-9 * mh + 1e-300;
    const double *__restrict__ slice = L.co + (size_t)i_inc * L.n_w * L.phi_pad;
    const float2 *__restrict__ blk = L.blk + (size_t)i_inc * L.nbr * L.nbc;
    const float2 *__restrict__ bnd = L.bandmm + (size_t)i_inc * L.nbands;
    w_lo = max(w_lo, 0); w_hi = min(max(w_hi, w_lo), L.n_w - 1); ip_lo = max(ip_lo, 0); ip_hi = min(max(ip_hi, ip_lo), L.n_phi - 1);
    const int br_lo = w_lo / R, br_hi = w_hi / R, bc_lo = ip_lo / C, bc_hi = ip_hi / C, ncb = bc_hi - bc_lo + 1;
    const int G = L.blk_g, tb_lo = br_lo / G, tb_hi = br_hi / G;
    const int lr = lane / C, lc = lane - lr * C;  // this lane's candidate inside a block
    double jub = jub_in;
    double best = inf, second = inf;  // pass 0: screening scores; pass 1: best = exact score of the lexicographic minimum
    int bflat = 0x7fffffff;
    double T = inf;  // pass 1: candidates with a screening score <= T are re-scored exactly
    int result = -1;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        unsigned nsw = 0;
#pragma unroll 1
        for (int t0 = tb_lo; t0 <= tb_hi; t0 += 64) {  // level 1: one lane per band
            const int t = t0 + lane;
            const bool tv = t <= tb_hi;
            double lb1 = inf;
            {
                const int tc = min(t, L.nbands - 1);
                const float2 mm = bnd[tc];
                const int r0 = tc * G * R, r1 = min((tc + 1) * G * R, L.n_w) - 1;
                const double wha = fma((double)r0, whs, wh0), whb = fma((double)r1, whs, wh0);
                const double dsg = fmax(0.0, fmax((double)mm.x - s, s - (double)mm.y)) * ainv;
                const double rad = fmax(0.0, fmax(wha - mh, mh - whb));
                lb1 = tv ? fma(dsg, dsg, rad * rad) : inf;
            }
            unsigned long long todo = __ballot(tv && !(lb1 * (1.0 - 1e-8) > jub + slack));
            if (todo == 0ULL) continue;
            int first = -1;
            if (pass == 0) {  // the most promising band first: its best candidate tightens the bound for all the others
                const double mn = wave_min_d(tv ? lb1 : 1e308);
                first = __ffsll((long long)(__ballot(tv && lb1 == mn) & todo)) - 1;
            }
            while (todo) {
                const int l = first >= 0 ? first : __ffsll((long long)todo) - 1;
                first = -1;
                todo &= ~(1ULL << l);
                if (rd_lane_d(lb1, l) * (1.0 - 1e-8) > jub + slack) continue;  // the bound has tightened since the ballot
                const int tband = t0 + l;
                const int brA = max(tband * G, br_lo), brB = min(min((tband + 1) * G, L.nbr), br_hi + 1);  // block rows [brA, brB)
                const int nblk = (brB - brA) * ncb;
#pragma unroll 1
                for (int k0 = 0; k0 < nblk; k0 += 64) {  // level 2: one lane per block of the band (inside the window)
                    const int idx = k0 + lane;
                    const bool bv = idx < nblk;
                    const int dr = bv ? idx / ncb : 0, dc = bv ? idx - dr * ncb : 0;
                    const int br = brA + dr, bc = bc_lo + dc;
                    const int r0 = br * R, r1 = min(r0 + R, L.n_w) - 1, c0 = bc * C, c1 = min(c0 + C, L.n_phi) - 1;
                    const float2 mm = blk[br * L.nbc + bc];
                    const double wha = fma((double)r0, whs, wh0), whb = fma((double)r1, whs, wh0);
                    const double dsg = fmax(0.0, fmax((double)mm.x - s, s - (double)mm.y)) * ainv;
                    const double rad = fmax(0.0, fmax(wha - mh, mh - whb));
                    double lbw = rad * rad;
                    if (L.blk_span_ok) {
                        const double2 ea = ((const double2 *)L.csphi)[c0], eb = ((const double2 *)L.csphi)[c1];
                        const bool inside = (ea.x * bh - ea.y * ah >= -tol) && (ah * eb.y - bh * eb.x >= -tol);
                        const double pmx = fmax(ah * ea.x + bh * ea.y, ah * eb.x + bh * eb.y);
                        const double tt = fmin(fmax(pmx, wha), whb);
                        const double e2 = m2 + tt * (tt - 2.0 * pmx);
                        lbw = inside ? lbw : fmax(lbw, e2);
                    }
                    const double lb2 = fma(dsg, dsg, lbw);
                    unsigned long long kept = __ballot(bv && !(lb2 * (1.0 - 1e-8) > jub + slack));
                    while (kept) {
                        const int kl = __ffsll((long long)kept) - 1;
                        kept &= kept - 1;
                        if (rd_lane_d(lb2, kl) * (1.0 - 1e-8) > jub + slack) continue;
                        // level 3: the block's 64 candidates, one per lane
                        const int row = rd_lane_i(br, kl) * R + lr, dir = rd_lane_i(bc, kl) * C + lc;
                        const bool ok = row < L.n_w && dir < L.n_phi;
                        const int rowc = min(row, L.n_w - 1), dirc = min(dir, L.n_phi - 1);
                        const double v = slice[(size_t)rowc * L.phi_pad + dirc];
                        const double2 cs = ((const double2 *)L.csphi)[dirc];
                        const double U = 2.0 * (ah * cs.x + bh * cs.y);
                        const double wh = fma((double)rowc, whs, wh0);
                        const double dd = fma(v, inv_dsig, sn);
                        double J = fma(dd, dd, wh * (wh - U));
                        J = ok ? J : inf;
                        const int flat = rowc * L.n_phi + dirc;
                        ++nsw;
                        if (pass == 0) {
                            second = vmin(second, vmax(J, best));
                            bflat = J < best ? flat : bflat;
                            best = vmin(best, J);
                            if (nsw <= 2u || (nsw & 3u) == 0u) {
                                const double g = wave_min_d(best);
                                if (g < 1e300) jub = fmin(jub, (g + m2) * (1.0 + 1e-9) + 1e-9);
                            }
                        } else if (J <= T) {  // exact re-scoring, numpy's first-minimum rule
                            const double Je = exact_J_co(L.w[rowc], cs.x, cs.y, v, s, a, b, dsig);
                            if (Je < best || (Je == best && flat < bflat)) { best = Je; bflat = flat; }
                        }
                    }
                    if (pass == 0 && nsw) {
                        const double g = wave_min_d(best);
                        if (g < 1e300) jub = fmin(jub, (g + m2) * (1.0 + 1e-9) + 1e-9);
                    }
                }
            }
        }
        cand += nsw * 64u;
        if (pass == 0) {
            const double gmin = wave_min_d(best);
            if (!(gmin < 1e300)) break;  // nothing scored (cannot happen: jub_in bounds a real candidate): exact scan below
            T = gmin + 1e-9 * (1.0 + fabs(gmin) + m2);
            const unsigned long long amb = __ballot(second <= T), surv = __ballot(best <= T);
            if (amb == 0ULL && __popcll(surv) == 1) { result = rd_lane_i(bflat, __ffsll((long long)surv) - 1); break; }
            // a near-tie: every candidate within eps of the minimum is re-scored in the reference's operation order (pass 1)
            jub = (T + m2) * (1.0 + 1e-9) + 1e-9;
            best = inf;
            bflat = 0x7fffffff;
        } else {
            wave_argmin(best, bflat);
            if (bflat != 0x7fffffff) result = bflat;
        }
    }
    if (result < 0) {
        went_exact = true;
        result = exact_scan_co(L, i_inc, s, a, b, dsig, lane);
    }
    return result;
}

// Stage 2 for narrow windows, `co_seg_pass<S>`: the wave takes 64/S pixels at a time, one per segment of S lanes
// (S = 16: windows of <= 16 directions, S = 32: <= 32).  What is wave-uniform in co_box_search is per-lane here,
// fetched once from the owner lanes (ds_bpermute), so the fixed cost of a pass -- lane layout, U, forward-difference
// set-up, argmin, settle -- is shared by 4 or 2 pixels.  All segments run the trip count of the longest; a segment
// that has covered its window keeps scoring the rows that follow (real candidates: harmless), its start slid down
// so that it never leaves the axis.  Pixels the pass cannot decide (near-ties, several survivors, trip count
// exceeding the axis) are returned in `redo` for co_box_search.
template <int S> __device__ __forceinline__ double seg_min_d(double v)
{
    v = vmin(v, dpp_d<0xB1, 0xF>(v));   // lane ^ 1
    if (S >= 4) v = vmin(v, dpp_d<0x4E, 0xF>(v));   // lane ^ 2
    if (S >= 8) v = vmin(v, dpp_d<0x141, 0xF>(v));  // 8-lane halves mirrored: every lane of an 8-lane group holds its minimum
    if (S >= 16) v = vmin(v, dpp_d<0x140, 0xF>(v));  // 16-lane rows mirrored
    if (S == 32) v = vmin(v, __shfl_xor(v, 16));     // (co_band_pass's 32-lane classes)
    return v;
}
template <int S>
__device__ __forceinline__ void co_seg_pass(const DevTables &L, const Pixel &P, const CoWindow &W, double inv_dsig, int lane,
                                            unsigned long long &pend, int &my_flat, unsigned long long &redo)
{
    constexpr int NP = 64 / S;
    const double BIG = 1e300;
    int o[NP], t_max = 0;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        o[q] = pend ? (__ffsll((long long)pend) - 1) : -1;
        if (pend) pend &= pend - 1;
        if (o[q] >= 0) t_max = max(t_max, rd_lane_i(W.geom, o[q]) >> 8);
    }
    const int q = lane / S, sl = lane & (S - 1);
    int own = o[0];
#pragma unroll
    for (int k = 1; k < NP; ++k) own = (q == k) ? o[k] : own;
    const bool valid = own >= 0;
    const int src = valid ? own : 0;
    const double s = __shfl(P.s_co, src), a = __shfl(P.a_re, src), b = __shfl(P.b_eff, src);
    const int i_inc = __shfl(P.i_inc, src), w_lo = __shfl(W.w_lo, src), ip_lo = __shfl(W.ip_lo, src);
    const int ncols = __shfl(W.ip_hi - W.ip_lo + 1, src), geom = __shfl(W.geom, src), mdiv = __shfl(W.mdiv, src);

    const double ah = 0.5 * a, bh = 0.5 * b, m2 = ah * ah + bh * bh, sn = -s * inv_dsig;
    const double wh0 = 0.5 * L.w0, whs = L.wstep_half;
    const int G = geom & 0xff;
    const int grp = (sl * mdiv) >> 16, col = sl - grp * ncols;
    const int rows_pass = t_max * 2 * G;
    const bool fits = rows_pass <= L.n_w && t_max <= XSW_MAX_FD_TRIPS;
    const bool act = valid && fits && grp < G;
    const int w_base = min(w_lo, L.n_w - rows_pass);
    const int row0 = act ? w_base + grp : 0, ip = act ? ip_lo + col : 0;
    const unsigned ipB = (unsigned)ip * 8u;
    const double U = 2.0 * (ah * *(const double *)((const char *)L.cphi + ipB) + bh * *(const double *)((const char *)L.sphi + ipB));
    const double dG = (double)G * whs;
    const double wh = fma((double)row0, whs, wh0);
    double pw = wh * (wh - U), dp = dG * (2.0 * wh - U) + dG * dG;
    const double snl = act ? sn : 1e150;  // idle lanes: dd ~ 1e150, score ~ 1e300 (finite: tags stay legal)
    const double ddp = 2.0 * dG * dG;
    const char *__restrict__ base = (const char *)L.co;
    const unsigned pstepB = (unsigned)(G * L.phi_pad) * 8u;
    unsigned off0 = (act ? ((unsigned)(i_inc * L.n_w + row0) * (unsigned)L.phi_pad + (unsigned)ip) : 0u) * 8u;
    unsigned off1 = off0 + (act ? pstepB : 0u);
    const unsigned adv = act ? 2u * pstepB : 0u;
    int keep_mask;
    asm("v_mov_b32 %0, 0xffff0000" : "=v"(keep_mask));
    double best = __builtin_inf(), second = __builtin_inf();
    double v0 = *(const double *)(base + off0), v1 = *(const double *)(base + off1);
    for (int t = 0; t < t_max; ++t) {
        off0 += adv;
        off1 += adv;
        double n0 = v0, n1 = v1;
        if (t + 1 < t_max) { n0 = *(const double *)(base + off0); n1 = *(const double *)(base + off1); }  // wave-uniform
        const double v[2] = {v0, v1};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double dd = fma(v[k], inv_dsig, snl);
            const double J = tag16(fma(dd, dd, pw), keep_mask, 2 * t + k);  // the score carries its step
            second = vmin(second, vmax(J, best));
            best = vmin(best, J);
            pw += dp;
            dp += ddp;
        }
        v0 = n0;
        v1 = n1;
    }
    const int row = row0 + (__double2loint(best) & 0xffff) * G;
    const double gmin = seg_min_d<S>(best);
    const double T = gmin + 1e-9 * (1.0 + fabs(gmin) + m2);
    const unsigned long long amb = __ballot(act && second <= T), surv = __ballot(act && best <= T);
    const unsigned long long bad = __ballot(valid && !fits);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if (o[k] < 0) continue;
        const unsigned long long bits = (S == 64 ? ~0ULL : ((1ULL << (S & 63)) - 1ULL)) << (k * S);
        const unsigned long long sv = surv & bits;
        if ((bad & bits) || (amb & bits) || __popcll(sv) != 1 || !(rd_lane_d(gmin, k * S) < 0.5 * BIG)) {
            redo |= 1ULL << o[k];
        } else {
            const int wl = __ffsll((long long)sv) - 1;
            const int flat = rd_lane_i(row, wl) * L.n_phi + rd_lane_i(ip, wl);
            if (lane == o[k]) my_flat = flat;
        }
    }
}

// Cross-pol search with exact interval pruning, one pixel per lane.  Precondition (host-checked, cr_monotone):
// every LUT row is non-decreasing in wind speed and the speed axis is uniform.  J = Jsig + Jwind with both
// terms >= 0, so for any candidate k0, J_ub = J(k0), no candidate with Jsig > J_ub can be the argmin:
// |LUT[k] - s| <= |dsig| sqrt(J_ub) is necessary, which on a monotone row is the index interval
// [lower_bound(s - d), upper_bound(s + d)) -- a handful of candidates when the cross-pol SNR is high.  k0 = the
// row entries bracketing s and the speed nearest to |wind_co|.  The lanes then sweep their intervals together
// (trip count = the longest interval in the wave).  Returns false (nothing done) when some lane's interval is
// long: the caller then runs the full-axis sweep for the whole wave.
__device__ __forceinline__ int lower_bound_row(const double *__restrict__ row, int n, double x)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (row[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ bool search_cr_interval(const DevTables &L, bool need, int i_inc, double s, double dsig,
                                                   bool have_co, double aco, int &icr, bool &undecided)
{
    const double inf = __builtin_inf();
    const double inv = 1.0 / dsig;
    const bool fast = need && L.cr_finite && isfinite(inv) && isfinite(s) && (!have_co || isfinite(aco));
    const double *__restrict__ row = L.cr + (size_t)(fast ? i_inc : 0) * L.wcr_pad;
    const double sn = fast ? -s * inv : 0.0, invf = fast ? inv : 0.0;
    const double f = (fast && have_co) ? 1.0 : 0.0;
    const double g = (fast && have_co) ? -0.5 * aco : 0.0;
    const int n = L.n_wcr;
    // w/2 of candidate k from the (uniform: cr_monotone implies it) axis instead of a table load; the screening tolerance
    // below covers the last-bit difference to the tabulated 0.5 * w[k]
    const double whs = L.wcrstep_half, wh0 = 0.5 * L.wcr0;
    auto score = [&](int k) {
        const double dd = fma(row[k], invf, sn);
        const double t = fma(fma((double)k, whs, wh0), f, g);
        return fma(t, t, dd * dd);
    };
    // upper bound from three candidates
    const double sq = fast ? s : 0.0;
    const int kb = lower_bound_row(row, n, sq);
    const int ka = min(max(kb, 0), n - 1), kp = min(max(kb - 1, 0), n - 1);
    const int kw = min(max((int)rint(((have_co ? aco : 0.0) - L.wcr0) * L.inv_wcrstep), 0), n - 1);
    double jub = vmin(score(ka), vmin(score(kp), score(have_co ? kw : ka)));
    jub = jub * (1.0 + 1e-9) + 1e-300;
    const double d = fabs(fast ? dsig : 1.0) * sqrt(jub) * (1.0 + 1e-9) + 1e-12 * (1.0 + fabs(sq));
    // the wind term is >= 0 too: ((w - |co|)/2)^2 <= J_ub, i.e. |w - |co|| <= 2 sqrt(J_ub): the index window [a0, a1] (with a
    // margin; uniform axis to 1e-12: csrc/xsw.hip uniform_axis).  The sigma0 interval is then looked for inside it only
    // (shorter bisections); the three seed candidates satisfy both bounds.
    int a0 = 0, a1 = n - 1;
    if (fast && have_co) {
        const double rw = 2.0 * sqrt(jub) * (1.0 + 1e-9) + 1e-9;
        const double xl = (aco - rw - L.wcr0) * L.inv_wcrstep, xh = (aco + rw - L.wcr0) * L.inv_wcrstep;
        const double nn = (double)n;
        a0 = max((int)ceil(fmin(fmax(xl - 1e-5 - 1e-9 * fabs(xl), -4.0), nn + 4.0)), 0);
        a1 = min((int)floor(fmin(fmax(xh + 1e-5 + 1e-9 * fabs(xh), -4.0), nn + 4.0)), n - 1);
        if (a1 < a0) { a0 = 0; a1 = n - 1; }  // cannot happen (the best seed lies inside); stay safe
    }
    const int span = a1 - a0 + 1;
    int lo = max(a0 + lower_bound_row(row + a0, span, sq - d) - 1, a0);
    int hi = min(a0 + lower_bound_row(row + a0, span, sq + d * (1.0 + 1e-15) + 1e-300) + 1, a1);  // >= upper_bound(s + d) within the window
    while (hi < a1 && row[hi] <= sq + d) ++hi;  // plateau at exactly s + d (never more than a step or two)
    const int len = fast ? (hi - lo + 1) : 0;
    if (__ballot(len > 160) != 0ULL) return false;  // a long interval somewhere in the wave: full sweep instead
    int maxlen = len;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off));
    double best = inf, second = inf;
    int code = 0;
    for (int t = 0; t < maxlen; ++t) {
        const bool ok = t < len;
        const int k = ok ? lo + t : lo;
        double J = score(k);
        J = ok ? J : inf;
        second = vmin(second, vmax(J, best));
        code = J < best ? k : code;
        best = vmin(best, J);
    }
    const double T = best + 1e-9 * (1.0 + fabs(best));
    icr = code;
    undecided = need && (!fast || !(best < inf) || second <= T);
    return true;
}

// Cross-pol search, one pixel per lane (all 64 pixels of the strip at once).  The speed axis has the same
// length for every pixel, so the trip count is wave-uniform and nothing diverges; lanes of one incidence
// bin read the same LUT word (one cache line per wave load).  Each lane sees ALL candidates of its pixel, so
// (best, second best) decide uniqueness directly; a lane that cannot decide (near-tie, non-finite input,
// non-finite LUT) reports `undecided` and is settled by the wave-cooperative exact scan.
__device__ __forceinline__ void search_cr_lanes(const DevTables &L, bool need, int i_inc, double s, double dsig,
                                                bool have_co, double aco, int &icr, bool &undecided)
{
    const double inf = __builtin_inf();
    const double inv = 1.0 / dsig;
    const bool fast = need && L.cr_finite && isfinite(inv) && isfinite(s) && (!have_co || isfinite(aco));
    const double *__restrict__ row = L.cr + (size_t)(fast ? i_inc : 0) * L.wcr_pad;
    const double sn = fast ? -s * inv : 0.0, invf = fast ? inv : 0.0;
    const double f = (fast && have_co) ? 1.0 : 0.0;   // Jwind_cr only when a co-pol wind exists (windspeed.py:259-264)
    const double g = (fast && have_co) ? -0.5 * aco : 0.0;
    double b4[4] = {inf, inf, inf, inf}, s4[4] = {inf, inf, inf, inf};
    int c4[4] = {0, 0, 0, 0};
    int k = 0;
    for (; k + 4 <= L.n_wcr; k += 4) {
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = row[k + q];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double dd = fma(v[q], invf, sn);
            const double t = fma(L.wcrh[k + q], f, g);   // (w - |co|)/2, wave-uniform table word
            const double J = fma(t, t, dd * dd);
            s4[q] = vmin(s4[q], vmax(J, b4[q]));
            const bool lt = J < b4[q];
            b4[q] = lt ? J : b4[q];
            c4[q] = lt ? (k + q) : c4[q];
        }
    }
    double best = inf, second = inf;
    int code = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        second = vmin(vmin(second, s4[q]), vmax(best, b4[q]));
        if (b4[q] < best) { best = b4[q]; code = c4[q]; }
    }
    for (; k < L.n_wcr; ++k) {
        const double dd = fma(row[k], invf, sn);
        const double t = fma(L.wcrh[k], f, g);
        const double J = fma(t, t, dd * dd);
        second = vmin(second, vmax(J, best));
        const bool lt = J < best;
        best = lt ? J : best;
        code = lt ? k : code;
    }
    const double T = best + 1e-9 * (1.0 + fabs(best));
    icr = code;
    undecided = need && (!fast || !(best < inf) || second <= T);
}

// hypot as glibc >= 2.35 computes it without FMA (sysdeps/ieee754/dbl-64/e_hypot.c, after C. Borges,
// "An improved algorithm for hypot(a,b)"): correctly rounded in all but very rare cases.  Verified
// bit-identical to numpy.hypot on 2e6 random pairs (tests/test_oracle.py).  Used for |wind_dual| in
// the fused dual-pol select.
__device__ __forceinline__ double hypot_glibc(double x, double y)
{
    if (!isfinite(x) || !isfinite(y)) return (isinf(x) || isinf(y)) ? __builtin_inf() : x + y;
    x = fabs(x); y = fabs(y);
    const double ax = x < y ? y : x, ay = x < y ? x : y;
    if (ax >= ay * 0x1p54) return ax + ay;
    if (ax > 0x1p511 || ay < 0x1p-459) return hypot(ax, ay);  // far outside any wind value
    double h = sqrt(ax * ax + ay * ay), t1, t2;
    if (h <= 2.0 * ay) {
        const double delta = h - ay;
        t1 = ax * (2.0 * delta - ax);
        t2 = (delta - 2.0 * (ax - ay)) * delta;
    } else {
        const double delta = h - ax;
        t1 = 2.0 * delta * (ax - 2.0 * ay);
        t2 = (4.0 * delta - ay) * ay + delta * delta;
    }
    h -= (t1 + t2) / (2.0 * h);
    return h;
}

// The interval rule of search_cr_interval without its bisections (L.inv_cr present): seeds are the rows around the tabulated
// crossing LUT ~ sigma0 plus the row nearest to |wind_co|; the best seed is admissible by construction (its score IS the
// bound), and the admissible set -- window |w - |co|| <= 2 sqrt(J_ub) AND band |LUT - s| <= |dsig| sqrt(J_ub) -- is an
// interval of the monotone row: it is scanned from the best seed upwards, then downwards, until the first inadmissible row on
// either side (one load per trip, trip count = the longest interval in the wave + 2).  Always returns true; an interval longer
// than XSW_CR_SCAN_MAX leaves the pixel undecided (exact scan by the caller).
#ifndef XSW_CR_SCAN_MAX
#define XSW_CR_SCAN_MAX 176
#endif
__device__ __forceinline__ bool search_cr_scan(const DevTables &L, bool need, int i_inc, double s, double dsig,
                                               bool have_co, double aco, int &icr, bool &undecided)
{
    const double inf = __builtin_inf();
    const double inv = 1.0 / dsig;
    const bool fast = need && L.cr_finite && isfinite(inv) && isfinite(s) && (!have_co || isfinite(aco));
    const bool windy = fast && have_co;  // the wind term ((w - |co|) / 2)^2 takes part
    const int ii = fast ? i_inc : 0;
    const double *__restrict__ row = L.cr + (size_t)ii * L.wcr_pad;
    const double sn = fast ? -s * inv : 0.0, invf = fast ? inv : 0.0;
    const double hco = windy ? 0.5 * aco : 0.0;
    const int n = L.n_wcr;
    const double whs = L.wcrstep_half, wh0 = 0.5 * L.wcr0;
    auto score_v = [&](int k, double v) {
        const double dd = fma(v, invf, sn);
        const double t = windy ? fma((double)k, whs, wh0) - hco : 0.0;
        return fma(t, t, dd * dd);
    };
    const double sq = fast ? s : 0.0;
    // seeds: three rows around the tabulated crossing, and the row nearest to |wind_co|
    const double *gr = L.inv_cr_grid + 3 * ii;
    const int bin = (int)fmin(fmax((sq - gr[0]) * gr[2], 0.0), (double)(XSW_INV_BINS - 1));
    const int k0 = (int)L.inv_cr[(size_t)ii * XSW_INV_BINS + bin];
    const int kw = min(max((int)rint(((have_co ? aco : 0.0) - L.wcr0) * L.inv_wcrstep), 0), n - 1);
    double jbest = inf;
    int ks = 0;
#pragma unroll
    for (int j = -1; j <= 2; ++j) {
        const int k = j == 2 ? (have_co ? kw : min(max(k0, 0), n - 1)) : min(max(k0 + j, 0), n - 1);
        const double J = score_v(k, row[k]);
        ks = J < jbest ? k : ks;
        jbest = vmin(jbest, J);
    }
    const double jub = jbest * (1.0 + 1e-9) + 1e-300;
    const double sj = sqrt(jub);
    const double d = fabs(fast ? dsig : 1.0) * sj * (1.0 + 1e-9) + 1e-12 * (1.0 + fabs(sq));
    int a0 = 0, a1 = n - 1;
    if (fast && have_co) {  // |w - |co|| <= 2 sqrt(J_ub), as an index window with a margin (see search_cr_interval)
        const double rw = 2.0 * sj * (1.0 + 1e-9) + 1e-9;
        const double xl = (aco - rw - L.wcr0) * L.inv_wcrstep, xh = (aco + rw - L.wcr0) * L.inv_wcrstep;
        const double nn = (double)n;
        a0 = max((int)ceil(fmin(fmax(xl - 1e-5 - 1e-9 * fabs(xl), -4.0), nn + 4.0)), 0);
        a1 = min((int)floor(fmin(fmax(xh + 1e-5 + 1e-9 * fabs(xh), -4.0), nn + 4.0)), n - 1);
    }
    a0 = min(a0, ks); a1 = max(a1, ks);  // the best seed is admissible whatever the rounding of the window did
    const double t_hi = sq + d, t_lo = sq - d;
    double best = jbest, second = inf;
    int code = ks, ku = ks + 1, kd = ks - 1;
    bool up = true, alive = fast;
#pragma unroll 1
    for (int t = 0; t < XSW_CR_SCAN_MAX; ++t) {
        if (__ballot(alive) == 0ULL) break;
        const int k = up ? ku : kd, kc = min(max(k, 0), n - 1);
        const double v = row[kc];
        const bool ok = alive && (up ? (k <= a1 && v <= t_hi) : (k >= a0 && v >= t_lo));
        double J = score_v(kc, v);
        J = ok ? J : inf;
        second = vmin(second, vmax(J, best));
        code = J < best ? kc : code;
        best = vmin(best, J);
        ku += (ok && up) ? 1 : 0;
        kd -= (ok && !up) ? 1 : 0;
        alive = alive && (ok || up);  // first miss going up: turn round; first miss going down: done
        up = up && ok;
    }
    const double T = best + 1e-9 * (1.0 + fabs(best));
    icr = code;
    undecided = need && (!fast || alive || !(best < inf) || second <= T);
    return true;
}

// numpy's complex128 true-divide (Smith) followed by np.angle, for the +phi / -phi choice
// (windspeed.py:236-242).
__device__ __forceinline__ double angle_of_quotient(double ar, double ai, double br, double bi)
{
    double qr, qi;
    const double abr = fabs(br), abi = fabs(bi);
    if (abr >= abi) {
        if (abr == 0.0 && abi == 0.0) { qr = ar / abr; qi = ai / abi; }
        else {
            const double rat = bi / br, scl = 1.0 / (br + bi * rat);
            qr = (ar + ai * rat) * scl;
            qi = (ai - ar * rat) * scl;
        }
    } else {
        const double rat = br / bi, scl = 1.0 / (bi + br * rat);
        qr = (ar * rat + ai) * scl;
        qi = (ai * rat - ar) * scl;
    }
    return atan2(qi, qr);
}

template <typename T> struct Cx;
template <> struct Cx<float> { typedef float2 type; };
template <> struct Cx<double> { typedef double2 type; };

template <typename T> __device__ __forceinline__ double ld(const void *p, long long i) { return (double)((const T *)p)[i]; }

// ------------------------------------------------------------------------------------------------

// Loads pixel `il` (already clamped in range), converts to dB, classifies it (windspeed.py:198-209,
// :252) and finds its incidence bins (:212, :254).
template <typename T, bool CR = true>
__device__ __forceinline__ void load_pixel(const DevTables &L, const KArgs &A, long long il, bool in, Pixel &P)
{
    const double nan = __builtin_nan("");
    const double inc = ld<T>(A.inc, il);
    P.s_co = nan; P.s_cr = nan; P.dsig = nan; P.a_re = nan; P.a_im = nan;
    if (A.s_co) P.s_co = to_db(((const T *)A.s_co)[il], A.is_db);
    if (CR && A.s_cr) {
        const T x = ((const T *)A.s_cr)[il];
        P.s_cr = to_db(x, A.is_db);
        // scalar dsig_cr is broadcast as sigma0_cr*0 + dsig_cr in the raster dtype (windspeed.py:122-123)
        P.dsig = A.dsig_cr ? (double)((const T *)A.dsig_cr)[il] : (double)(T)(x * (T)0 + (T)A.dsig_cr_scalar);
    }
    if (A.anc) {
        typename Cx<T>::type z = ((const typename Cx<T>::type *)A.anc)[il];
        P.a_re = (double)z.x;
        P.a_im = (double)z.y;
    }
    P.flags = 0; P.i_inc = 0; P.i_inc_cr = 0;
    if (in) {
        const bool anc_nan = (P.a_re != P.a_re || P.a_im != P.a_im) && !(isinf(P.a_re) || isinf(P.a_im));  // isnan(hypot)
        if (inc != inc || (P.s_co == P.s_co && anc_nan)) P.flags = F_EARLY_NAN;   // windspeed.py:198-207
        else {
            if (P.s_co == P.s_co) {
                P.flags |= F_NEED_CO;
                P.i_inc = nearest_index(L.inc, L.n_inc, inc, L.inc_uniform != 0, L.inc0, L.inv_incstep);
                if (fabs(P.s_co) < 1e100 && fabs(P.a_re) < 1e100 && fabs(P.a_im) < 1e100) P.flags |= F_CO_FINITE;  // NaN/inf/absurd: exact scan
            }
            if (P.s_cr == P.s_cr && P.dsig == P.dsig) {
                P.flags |= F_NEED_CR;
                P.i_inc_cr = nearest_index(L.inc_cr, L.n_inc_cr, inc, L.inc_cr_uniform != 0, L.inc_cr0, L.inv_inccrstep);
            }
        }
    }
    P.b_eff = L.phi_180 ? fabs(P.a_im) : P.a_im;  // windspeed.py:218-219
    // search-window geometry of the branch-and-bound kernel, one pixel per lane (64 at a time)
    P.mag = (double)__builtin_sqrtf((float)(P.a_re * P.a_re + P.b_eff * P.b_eff));
    double th = (double)atan2f((float)P.b_eff, (float)P.a_re) * 57.295779513082320877;
    if (th < L.phi0) th += 360.0;
    P.theta = th;
    P.ipr = (P.flags & F_CO_FINITE)
                ? (int)rint(fmin(fmax((th - L.phi0) * L.inv_dphi, 0.0), (double)(L.n_phi - 1))) : 0;
}

// Forms pixel i's complex winds from the winning indices and stores them (windspeed.py:231-250,
// :269-281, dual select :426-428).
template <typename TO, bool CR = true>
__device__ __forceinline__ void store_pixel(const DevTables &L, const KArgs &A, long long i, const Pixel &P,
                                            int my_flat, int my_icr)
{
    const double nan = __builtin_nan("");
    double co_re, co_im, cr_re, cr_im;
    int o_iw = -1, o_ip = -1;
    int sgn = 0;
    if (P.flags & F_EARLY_NAN) {
        co_re = nan; co_im = 0.0; cr_re = nan; cr_im = 0.0;  // out[i] = np.nan -> (nan + 0j)
    } else {
        if (P.flags & F_NEED_CO) {
            // flat < 2^30, n_phi < 2^16: (flat + 0.5) / n_phi is at least 0.5 / n_phi away from an integer, the product's error ~1e-7 of that
            o_iw = (int)(((double)my_flat + 0.5) * L.inv_nphi);
            o_ip = my_flat - o_iw * L.n_phi;
            const double w = L.w[o_iw];
            const double2 e1 = ((const double2 *)L.out_dir)[o_ip];
            const double s1r = w * e1.x, s1i = w * e1.y + 0.0 * e1.x;  // float * complex (:235)
            co_re = s1r; co_im = s1i;
            if (L.phi_180) {
                const double2 e2 = ((const double2 *)L.out_dir)[L.n_phi + o_ip];
                const double s2r = w * e2.x, s2i = w * e2.y + 0.0 * e2.x;
                // |angle(a / s1)| <= |angle(a / s2)|  with  s1,2 = w e^{+-i phi}:  cos(angle1) - cos(angle2) = 2 sin(theta_a) sin(phi),
                // so away from sin(theta_a) sin(phi) = 0 the choice is the sign of Im(a) * sin(phi) (the computed angles are good
                // to ~1e-15; the margin asked for here is 1e-9).  Only the near-ties go through the emulated division + atan2.
                const double xs = P.a_im * e1.y, mag = fabs(P.a_re) + fabs(P.a_im);
                const bool clear = fabs(xs) > 1e-9 * mag && mag > 1e-100 && mag < 1e100 && w > 1e-100 && w < 1e100 && e2.y == -e1.y &&
                                   e2.x == e1.x;
                bool second = xs < 0.0;
                if (!clear) {
                    const double d1 = angle_of_quotient(P.a_re, P.a_im, s1r, s1i);
                    const double d2 = angle_of_quotient(P.a_re, P.a_im, s2r, s2i);
                    second = !(fabs(d1) <= fabs(d2));
                }
                if (second) { co_re = s2r; co_im = s2i; sgn = 1; }
            }
        } else {
            co_re = nan; co_im = nan;  // np.nan * 1j
        }
        if (P.flags & F_NEED_CR) {
            const double wd = L.wcr[my_icr];
            if (P.flags & F_NEED_CO) {  // |wind_co| is never NaN once a co-pol search ran
                const double2 u = ((const double2 *)L.dual_dir)[((size_t)sgn * L.n_w + o_iw) * L.n_phi + o_ip];
                cr_re = wd * u.x;
                cr_im = wd * u.y + 0.0 * u.x;
            } else {
                cr_re = wd; cr_im = 0.0;  // exp(1j*0)
            }
        } else {
            cr_re = nan; cr_im = nan;
        }
    }
    typedef typename Cx<TO>::type cx_t;
    if (A.out_co) {
        cx_t z; z.x = (TO)co_re; z.y = (TO)co_im;
        ((cx_t *)A.out_co)[i] = z;
    }
    // the answer as grid codes (xsw.h): what xsw_expand_codes turns back into exactly the values formed above
    if (A.code_co)
        A.code_co[i] = (P.flags & F_EARLY_NAN) ? K_CODE_NAN_RE : (P.flags & F_NEED_CO) ? ((unsigned)my_flat | ((unsigned)sgn << 30)) : K_CODE_NAN;
    if (CR && (A.out_cr || A.code_cr)) {
        bool picked_co = false;
        if (A.dual_select) {  // xr.where((|co| < 5) | (|dual| < 5), co, dual)  (windspeed.py:426-428)
            // without a co-pol search wind_co is (nan, nan) or (nan, 0): |wind_co| is NaN, never < 5
            const double aco = (P.flags & F_NEED_CO) ? L.abs_co[(size_t)o_iw * L.n_phi + o_ip] : nan;
            // |wind_dual| = wspd_dual * |unit vector| = wspd_dual (1 +- 1e-15): the emulated hypot only decides next to 5 m/s
            bool dual_small = false;  // no cross-pol search: (nan, nan)
            if (!(P.flags & F_EARLY_NAN) && (P.flags & F_NEED_CR)) {
                const double wd = L.wcr[my_icr];
                dual_small = wd < 5.0 - 1e-9 ? true : (wd > 5.0 + 1e-9 ? false : hypot_glibc(cr_re, cr_im) < 5.0);
            }
            if (aco < 5.0 || dual_small) { cr_re = co_re; cr_im = co_im; picked_co = true; }
        }
        if (A.out_cr) {
            cx_t z; z.x = (TO)cr_re; z.y = (TO)cr_im;
            ((cx_t *)A.out_cr)[i] = z;
        }
        if (A.code_cr)
            A.code_cr[i] = (P.flags & F_EARLY_NAN) ? K_CODE_NAN_RE
                                                     : (((P.flags & F_NEED_CR) ? (unsigned)my_icr : K_CODE_NO_INDEX) | (picked_co ? K_CODE_PICK_CO : 0u));
    }
    if (A.out_idx) {
        A.out_idx[3 * i + 0] = o_iw;
        A.out_idx[3 * i + 1] = o_ip;
        A.out_idx[3 * i + 2] = (P.flags & F_NEED_CR) ? my_icr : -1;
    }
}

// Production kernel.  ALGO: 1 = branch-and-bound, 3 = exact full sweep for every pixel.
// The unconstrained allocation is kept (mono instantiation 78 VGPRs = 6 waves per SIMD, dual-pol 88 = 5; no scratch):
// forcing 6 waves on the dual-pol kernel spills (+0.7 GB of HBM writes per 4e8-pixel launch), 8 spills in the sweep.
#ifndef XSW_INVERT_WAVES_PER_SIMD
#define XSW_INVERT_WAVES_PER_SIMD 1
#endif
// CR = false: mono co-pol instantiation (no cross-pol raster, no second output): the cross-pol pixel state is not kept
// alive through the co-pol search.
// Inverts the (up to) 64 pixels of one wave: lane l owns pixel i (in = the lane has one).  Body of k_invert / k_invert_list.
template <typename T, typename TO, int ALGO, bool CR>
__device__ __forceinline__ void invert_strip(const DevTables &L, const KArgs &A, long long i, bool in, int lane)
{
    const double nan = __builtin_nan("");

    Pixel P;
    load_pixel<T, CR>(L, A, i, in, P);

    // ---- co-pol search.  Stage 1 (one pixel per lane): upper bound + search window.  Stage 2: the wave
    //      walks its pixels one at a time, window and parameters wave-uniform (readlane -> SGPRs).
    int my_flat = -1, my_icr = -1;
    unsigned cand = 0, n_exact = 0, n_co = 0, n_cr = 0;
    const bool use_prune = ALGO == 1 && L.prunable;
    CoWindow W;
    W.w_lo = W.w_hi = W.ip_lo = W.ip_hi = W.geom = W.mdiv = 0;
    unsigned long long todo = __ballot((P.flags & F_NEED_CO) != 0);
    unsigned long long relay = 0;  // pixels laid out for a 16/32-lane segment that co_box_search has to lay out again
    if (use_prune && todo) {
        bool loose = false;
        W = co_window_lanes<XSW_STRIP_RAYS, 2>(L, P, A.inv_dsig_co, fabs(A.dsig_co), loose);
        if (loose) P.flags = (P.flags & ~F_CO_FINITE) | (L.blk ? F_CO_LOOSE : 0);  // no forward differences for it: block pyramid (exact scan without the tables)
        const unsigned long long fin_m = __ballot((P.flags & F_NEED_CO) != 0 && (P.flags & F_CO_FINITE) != 0);
        cand += (unsigned)__popcll(fin_m) * (unsigned)(2 * (32 - __clz((L.n_w + 1) >> 1)) + (XSW_STRIP_RAYS - 1) * 2 * XSW_RAY_SIDE_STEPS);
        if (L.co_off32) {
            const int ncols_p = W.ip_hi - W.ip_lo + 1;
            const bool elig = (P.flags & F_NEED_CO) != 0 && (P.flags & F_CO_FINITE) != 0;
            unsigned long long m4 = __ballot(XSW_SEG4 && elig && ncols_p <= 4);
            unsigned long long m8 = __ballot(XSW_SEG8 && elig && ncols_p <= 8) & ~m4;
            unsigned long long m16 = __ballot(elig && ncols_p <= 16) & ~(m8 | m4);
            const unsigned long long segd = m4 | m8 | m16;
            relay |= segd;
            if (A.stats) {
                unsigned c = ((segd >> lane) & 1ULL) ? (unsigned)((W.w_hi - W.w_lo + 1) * ncols_p) : 0u;
                for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
                cand += c;
            }
            unsigned long long redo = 0;
            while (m4) co_seg_pass<4>(L, P, W, A.inv_dsig_co, lane, m4, my_flat, redo);
            while (m8) co_seg_pass<8>(L, P, W, A.inv_dsig_co, lane, m8, my_flat, redo);
            while (m16) co_seg_pass<16>(L, P, W, A.inv_dsig_co, lane, m16, my_flat, redo);
            n_co += (unsigned)__popcll(segd & ~redo);
            todo = (todo & ~segd) | redo;
        }
    }
    while (todo) {
        const int p = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int uf = rd_lane_i(P.flags, p);
        const int u_iinc = rd_lane_i(P.i_inc, p);
        const double us = rd_lane_d(P.s_co, p), ua = rd_lane_d(P.a_re, p), ub = rd_lane_d(P.b_eff, p);
        int flat = -1;
        bool went_exact = false;
        if (use_prune && (uf & (F_CO_FINITE | F_CO_LOOSE))) {
            const int wl = rd_lane_i(W.w_lo, p), wh = rd_lane_i(W.w_hi, p), il = rd_lane_i(W.ip_lo, p), ih = rd_lane_i(W.ip_hi, p);
            // the block pyramid (bounds from the table side) takes the large windows, the pixels whose bound is loose, and what the
            // window sweep cannot settle (near-ties, more trips than its forward differences allow)
            bool blocks = L.blk != nullptr && ((uf & F_CO_LOOSE) != 0 || (long long)(wh - wl + 1) * (ih - il + 1) >= (long long)A.block_min);
            if (!blocks) {
                bool defer = false;
                flat = co_box_search(L, u_iinc, us, ua, ub, wl, wh, il, ih, rd_lane_i(W.geom, p), rd_lane_i(W.mdiv, p), A.dsig_co, A.inv_dsig_co, lane, cand,
                                     went_exact, ((relay >> p) & 1ULL) != 0, L.blk ? &defer : nullptr);
                blocks = defer;
            }
            if (blocks) {
                const double rs = rd_lane_d(W.band_d, p) * fabs(A.inv_dsig_co);  // >= sqrt(J_ub) (co_window_lanes)
                flat = co_block_search(L, u_iinc, us, ua, ub, rs * rs * (1.0 + 1e-12), wl, wh, il, ih, A.dsig_co, A.inv_dsig_co, lane, cand, went_exact);
            }
        } else {
            flat = exact_scan_co(L, u_iinc, us, ua, ub, A.dsig_co, lane);
            went_exact = true;
            cand += (unsigned)(L.n_w * L.n_phi);
        }
        n_exact += went_exact ? 1u : 0u;
        n_co += 1u;
        if (lane == p) my_flat = flat;
    }

    // ---- cross-pol search (windspeed.py:252-269): one pixel per lane, then the undecided ones cooperatively
    if (CR && A.s_cr) {
        const bool need_cr = (P.flags & F_NEED_CR) != 0;
        const bool have_co = (P.flags & F_NEED_CO) != 0;  // |wind_co| is never NaN once a co-pol search ran
        const double aco = have_co ? L.abs_co[my_flat] : nan;  // np.abs(wind_co), table [n_w][n_phi]
        bool undecided = need_cr;
        if (ALGO == 1) {
            bool done = false;
            if (L.cr_monotone && L.inv_cr) done = search_cr_scan(L, need_cr, P.i_inc_cr, P.s_cr, P.dsig, have_co, aco, my_icr, undecided);
            else if (L.cr_monotone) done = search_cr_interval(L, need_cr, P.i_inc_cr, P.s_cr, P.dsig, have_co, aco, my_icr, undecided);
            if (!done) search_cr_lanes(L, need_cr, P.i_inc_cr, P.s_cr, P.dsig, have_co, aco, my_icr, undecided);
        }
        n_cr = (unsigned)__popcll(__ballot(need_cr));
        unsigned long long und = __ballot(undecided);
        while (und) {
            const int p = __ffsll((long long)und) - 1;
            und &= und - 1;
            const int k = exact_scan_cr(L, rd_lane_i(P.i_inc_cr, p), rd_lane_d(P.s_cr, p), rd_lane_d(P.dsig, p),
                                        rd_lane_i((int)have_co, p) != 0, rd_lane_d(aco, p), lane);
            if (lane == p) my_icr = k;
        }
    }
    if (A.stats && lane == 0) {
        atomicAdd(&A.stats[0], (unsigned long long)n_co);
        atomicAdd(&A.stats[1], (unsigned long long)cand);
        if (A.stats_chain) atomicAdd(&A.stats[6], (unsigned long long)cand);
        atomicAdd(&A.stats[2], (unsigned long long)n_exact);
        atomicAdd(&A.stats[3], (unsigned long long)n_cr);
    }
    if (in) store_pixel<TO, CR>(L, A, i, P, my_flat, my_icr);
}

// Production kernel (general path).  ALGO: 1 = branch-and-bound, 3 = exact full sweep for every pixel.
// CR = false: mono co-pol instantiation (no cross-pol raster, no second output): the cross-pol pixel state is not kept
// alive through the co-pol search.
template <typename T, typename TO, int ALGO, bool CR = true>
__global__ __launch_bounds__(256, XSW_INVERT_WAVES_PER_SIMD) void k_invert(DevTables L, KArgs A)
{
    const int lane = threadIdx.x & 63;
    // Raster tile of this workgroup: 4 lines x 64 samples (one strip per wave).  Incidence varies along
    // `sample` only, so a column of tiles shares one or two LUT slices.  Workgroups are dealt round-robin
    // over the 8 XCDs (b % 8 shares an XCD; placement is a speed heuristic, never a correctness
    // assumption): XCD x walks its own contiguous range of tile columns, line groups fastest, so that the
    // waves resident on one XCD at any time work in the same few slices and its L2 keeps them.
    const long long strips_per_line = (A.samples + 63) >> 6, line_groups = (A.lines + 3) >> 2;
    const long long cols_per_xcd = (strips_per_line + 7) >> 3;
    const long long xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const long long col = xcd * cols_per_xcd + j / line_groups;
    const long long line = (j % line_groups) * 4 + (threadIdx.x >> 6);
    if (j / line_groups >= cols_per_xcd || col >= strips_per_line || line >= A.lines) return;  // wave-uniform
    const long long smp = col * 64 + lane;
    const bool in = smp < A.samples;
    const long long i = line * A.samples + (in ? smp : A.samples - 1);
    invert_strip<T, TO, ALGO, CR>(L, A, i, in, lane);
}

// Second kernel of the two-kernel path: inverts the pixels k_invert_band left undecided (A.list, A.list_count), 64 per wave,
// with the general algorithm.  The pixels are scattered, so their rasters are gathered; they are few (0.2 % on the
// benchmark scene).  Fixed grid; every wave strides over the list.
// The compiler's free choice for this kernel is 152 VGPRs (175 dual-pol) = 3 (2) waves per SIMD, and the kernel waits for loads
// 68 % of the time (rocprofv3, a-priori x 1.6 scene: 5400 VALU wave-instructions and 580 loads per listed pixel, VALU issue 28 %).
// Capped at 128 VGPRs = 4 waves per SIMD (20 B of spills): list time 562 -> 500 ms on that scene, 19 -> 15 ms on a-priori x 0.6,
// 18.8 -> 16.0 ms on incidence 17..25 deg; 5 / 6 / 8 waves spill 130-370 B and are faster on some scenes, slower on others.
#ifndef XSW_LIST_WAVES
#define XSW_LIST_WAVES 4
#endif
template <typename T, typename TO, bool CR>
__global__ __launch_bounds__(256, XSW_LIST_WAVES) void k_invert_list(DevTables L, KArgs A)
{
    const int lane = threadIdx.x & 63;
    const long long count = (long long)*A.list_count;
    const long long nwaves = (long long)gridDim.x * 4;
    const long long strips_per_line = (A.samples + 63) >> 6, line_groups = (A.lines + 3) >> 2;
    const long long cols_per_xcd = (strips_per_line + 7) >> 3, nb = 8 * cols_per_xcd * line_groups;
    if (count > (long long)A.list_cap && !A.mask_g) {
        // the list overflowed (k_invert_band kept counting but could not append) and there are no strip masks: its pixels are
        // unknown, so every tile of the raster is inverted by the general algorithm -- k_invert's tile walk as a grid-stride loop.
        // Results do not depend on which kernel wrote a pixel.
        for (long long b = blockIdx.x; b < nb; b += gridDim.x) {
            const long long xcd = b & 7, j = b >> 3;
            const long long col = xcd * cols_per_xcd + j / line_groups;
            const long long line = (j % line_groups) * 4 + (threadIdx.x >> 6);
            if (col >= strips_per_line || line >= A.lines) continue;  // wave-uniform
            const long long smp = col * 64 + lane;
            const bool in = smp < A.samples;
            invert_strip<T, TO, 1, CR>(L, A, line * A.samples + (in ? smp : A.samples - 1), in, lane);
        }
        return;
    }
    const long long nlist = count < (long long)A.list_cap ? count : (long long)A.list_cap;
    // XSW_LIST_PX pixels per wave and pass: the cooperative stage takes the pixels one after the other, so fewer pixels per
    // wave spread a short list over more SIMDs (the per-lane stage runs with idle lanes, which costs nothing here)
#ifndef XSW_LIST_PX
#define XSW_LIST_PX 16
#endif
    // a long list (a LUT or scene the band rule rarely applies to) fills the waves instead: 64 pixels per wave (measured at 2.3e5
    // pixels: 16 per wave 2.6 ms, 64 per wave 3.5 ms); a SHORT list is spread over all the waves of the grid -- 1, 2, 4, 8 pixels
    // per wave: what is left on a short list are the heaviest windows of the scene (1e4 candidates each), and a wave takes its
    // pixels one after the other, so 16 of them in one wave were the whole of this kernel's time on the benchmark scene
    int ppw = 64;
    if (nlist < 64LL * nwaves) {
        ppw = 1;
        while (ppw < XSW_LIST_PX && (long long)ppw * nwaves < nlist) ppw <<= 1;
    }
    for (long long c = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); c * ppw < nlist; c += nwaves) {  // wave-uniform
        const long long k = c * ppw + lane;
        const bool in = lane < ppw && k < nlist;
        const long long i = (long long)A.list[in ? k : nlist - 1];
        invert_strip<T, TO, 1, CR>(L, A, i, in, lane);
    }
    if (count > (long long)A.list_cap) {
        // the pixels that did not fit into the list are marked in mask_g (one bit per pixel, a word per strip): the marked pixels
        // of the marked strips, k_invert's tile walk as a grid-stride loop, rasters read in place
        for (long long b = blockIdx.x; b < nb; b += gridDim.x) {
            const long long xcd = b & 7, j = b >> 3;
            const long long col = xcd * cols_per_xcd + j / line_groups;
            const long long line = (j % line_groups) * 4 + (threadIdx.x >> 6);
            if (col >= strips_per_line || line >= A.lines) continue;  // wave-uniform
            const unsigned long long m = A.mask_g[line * strips_per_line + col];  // wave-uniform address
            if (m == 0ULL) continue;
            const long long smp = col * 64 + lane;
            const bool in = smp < A.samples;
            invert_strip<T, TO, 1, CR>(L, A, line * A.samples + (in ? smp : A.samples - 1), in && ((m >> lane) & 1ULL) != 0ULL, lane);
        }
    }
}

}  // namespace xsw
