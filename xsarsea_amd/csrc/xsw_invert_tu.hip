// The search kernels of ONE (input dtype, output dtype) pair and their launch logic (-DXSW_PAIR=0..3; xsarsea_amd/_build.py
// compiles the four side by side and links them with xsw.hip into libxsw.so).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <string>

#include "xsw_host.hpp"
#include "xsw_band.hpp"
#include "xsw_band2.hpp"
#include "xsw_blocks.hpp"
#include "xsw_exhaustive.hpp"

using namespace xsw;

#ifndef XSW_B2_AREA
#define XSW_B2_AREA 2048  // measured with list C at half the raster (profiles/sweep_b2_area.sh, Mpx/s at 1e6 / 8192 / 4096 / 2048 / 1024 / 512): outliers 5 % 727 / 2486 / 2675 / 2711 / 2624 / 2694, a-priori x 0.3 424 / 440 / 512 / 591 / 643 / 620, x 2.5 460 / 459 / 480 / 520 / 508 / 489, x 0.6 1148 / 1147 / 1161 / 1176 / 1128 / 910
#endif
#ifndef XSW_LONG_RUN_DEFAULT
#define XSW_LONG_RUN_DEFAULT 5
#endif
#ifndef XSW_ARC_MIN
#define XSW_ARC_MIN 32    // directions from which a window is narrowed to its live arc in stage 1 of k_invert_band (environment XSW_ARC_MIN; 0: never).  48 / 40 / 32 / 24 at 48 pixels per wave: a-priori x 1.6 1651 / 1663 / 1711 / 1736 Mpx/s, inc 17-33 x 1.6 1012 / 1059 / 1081 / 1071, cyclone band 7816 / 7885 / 7865 / 7637
#endif
#ifndef XSW_ARC_CROWD
#define XSW_ARC_CROWD 48  // ... when this many of the wave's 64 pixels are such (environment XSW_ARC_CROWD)
#endif
#ifndef XSW_B2_CROWD
#define XSW_B2_CROWD 24  // pixels beyond XSW_B2_AREA a wave of k_invert_band must hold (of 64) for them to stay k_invert_band2's (environment XSW_B2_CROWD; 65: never)
#endif
#ifndef XSW_B2_WIDE
#define XSW_B2_WIDE 0  // directions from which a window is k_invert_band2's whatever its run (0: never)
#endif
#ifndef XSW_B2_REFINE_MIN
#define XSW_B2_REFINE_MIN 16  // records marked for the refinement a wave of k_invert_band2 must hold to run it (environment XSW_B2_REFINE_MIN)
#endif
#ifndef XSW_BLOCK_MIN
#define XSW_BLOCK_MIN 1024
#endif

template <typename T, typename TO>
static int launch_invert(xsw_ctx *c, const KArgs &A_in, int algo, const LaunchCtl &lc, std::string &err)
{
    KArgs A = A_in;
    // k_invert grid: 8 XCD lanes x ceil(columns/8) tile columns x line groups (see the kernel)
    const long long strips_per_line = (A.samples + 63) / 64, line_groups = (A.lines + 3) / 4;
    const long long nblocks = 8 * ((strips_per_line + 7) / 8) * line_groups;
    if (nblocks > 0x7fffffffLL) return seterr(err, XSW_EINVAL, "raster too large for one launch");
    const bool mono = !A.s_cr && !A.out_cr && !A.code_cr;
    if (algo == XSW_ALGO_EXHAUSTIVE || algo == XSW_ALGO_EXHAUSTIVE_F64)
        return launch_exhaustive<T, TO>(c->T, A, lc.stream, algo == XSW_ALGO_EXHAUSTIVE) == hipSuccess
                   ? XSW_OK : seterr(err, XSW_EHIP, "exhaustive launch failed: %s", hipGetErrorString(hipGetLastError()));
    // Two-kernel fast path: k_invert_band finishes every pixel the band rule decides (monotone LUT rows, finite inputs,
    // unique minimum; cross-pol by the interval rule) and appends the rest to a work list; k_invert_list inverts those
    // (all tiles, should the list overflow).
    static const int block_min_env = getenv("XSW_BLOCK_MIN") ? std::max(0, atoi(getenv("XSW_BLOCK_MIN"))) : XSW_BLOCK_MIN;
    A.block_min = block_min_env;  // windows of at least this many candidates: block pyramid (general kernel)
    static const bool band_off = getenv("XSW_NO_BAND") != nullptr;  // experiments / A-B measurements only
    if (algo == XSW_ALGO_PRUNED && !band_off && lc.list && A.s_co && c->T.prunable && c->T.mono_rows && c->T.inv_rows && c->T.co_off32 && c->T.band_mul24 &&
        (!A.s_cr || c->T.cr_monotone) && A.n < (1LL << 32)) {
        KArgs B = A;
        B.list_count = lc.list;
        B.list = lc.list + 16;
        B.list_cap = (unsigned)std::min<size_t>(lc.list_cap, 0xfffffff0u);
        // list B (k_invert_band -> k_invert_band2) follows list G: a pixel whose band holds XSW_LONG_RUN (5) or more rows along the
        // a-priori direction is handed to k_invert_band2 -- one such pixel holds up every pixel of its pass in k_invert_band, and
        // where the a-priori wind is far from the sigma0 contour most pixels are such.  XSW_LONG_RUN=0: never (k_invert_band
        // sweeps every window: A/B measurements); the statistics instantiation sweeps every window in k_invert_band as well.
        // (5 since round 5: with the stage-1 live arc and the cheaper k_invert_band2 re-measured on the hard scenes, 4 / 5 / 6 / 8 rows: cyclone band
        // 7669 / 8052 / 7971 / 7489 Mpx/s, outliers 5 % 3640 / 3885 / 3955 / 3860, a-priori x 0.6 1925 / 2073 / 2081 / 1908, x 1.6 1586 / 1610 /
        // 1571 / 1480, inc 17-33 x 1.6 979 / 1009 / 1030 / 990; the 20000 x 20000 benchmark scene 37.31 / 37.44 / 37.77 ms: within its noise for 4 / 5)
        static const int long_run_env = getenv("XSW_LONG_RUN") ? std::max(0, atoi(getenv("XSW_LONG_RUN"))) : XSW_LONG_RUN_DEFAULT;
        const bool count_inst = A.stats && !A.stats_chain;  // the statistics instantiation: k_invert_band sweeps every window itself and counts
        const bool band2 = long_run_env > 0 && !count_inst;
        if (band2) { B.list_b_count = lc.list + 1; B.list_b = lc.list + 16 + lc.list_cap; B.list_b_cap = (unsigned)std::min<size_t>(XSW_LIST_B_SHARE * lc.list_cap, 0xfffffff0u); }
        static const bool records_off = getenv("XSW_NO_RECORDS") != nullptr;  // A/B measurements and the tests of the index-list route
        static_assert(sizeof(BandRec) == XSW_REC_BYTES, "XSW_REC_BYTES (xsw_host.hpp) is sizeof(BandRec)");
        B.rec_b = (band2 && !records_off) ? lc.rec_b : nullptr;
        // list C (k_invert_band -> k_invert_blocks): the finite pixels the band rule is not for.  XSW_NO_BLOCKS_KERNEL=1: they stay on
        // list G, i.e. with k_invert_list (A/B measurements and the tests of that route)
        static const bool blocks_kernel_off = getenv("XSW_NO_BLOCKS_KERNEL") != nullptr;
        const bool blocks3 = c->T.blk != nullptr && c->T.blk_span_ok && !blocks_kernel_off && c->T.n_w < 32768 && c->T.n_phi < 32768;
        if (blocks3) { B.list_c_count = lc.list + 2; B.list_c = lc.list + 16 + (1 + XSW_LIST_B_SHARE) * lc.list_cap; B.list_c_cap = (unsigned)std::min<size_t>(XSW_LIST_C_SHARE * lc.list_cap, 0xfffffff0u); }
        B.long_run = long_run_env;
        static const int area_max_env = getenv("XSW_B2_AREA") ? std::max(1, atoi(getenv("XSW_B2_AREA"))) : XSW_B2_AREA;
        B.area_max = c->T.blk ? area_max_env : 0x7fffffff;  // (without the block tables the general kernel has nothing better to offer)
        static const int crowd_env = getenv("XSW_B2_CROWD") ? std::max(1, atoi(getenv("XSW_B2_CROWD"))) : XSW_B2_CROWD;
        B.b2_crowd = crowd_env;  // (65: never)
        B.area_crowd_max = 1 << 20;
        static const int wide_env = getenv("XSW_B2_WIDE") ? std::max(0, atoi(getenv("XSW_B2_WIDE"))) : XSW_B2_WIDE;
        B.wide_min = wide_env > 0 ? wide_env : 0x7fffffff;
        static const int arc_min_env = getenv("XSW_ARC_MIN") ? atoi(getenv("XSW_ARC_MIN")) : XSW_ARC_MIN;
        static const int arc_crowd_env = getenv("XSW_ARC_CROWD") ? std::max(1, atoi(getenv("XSW_ARC_CROWD"))) : XSW_ARC_CROWD;
        B.arc_min = (arc_min_env > 0 && c->T.csphi32) ? arc_min_env : 0x7fffffff;
        B.arc_crowd = arc_crowd_env;
        static const int refine_min_env = getenv("XSW_B2_REFINE_MIN") ? std::max(0, atoi(getenv("XSW_B2_REFINE_MIN"))) : XSW_B2_REFINE_MIN;
        B.b2_refine_min = refine_min_env;
        static const int b2_rows_env = getenv("XSW_B2_ROWS_MAX") ? std::max(1, atoi(getenv("XSW_B2_ROWS_MAX"))) : XSW_B2_ROWS_MAX;
        B.b2_rows_max = b2_rows_env;
        static const int tail_max_env = getenv("XSW_TAIL_SWEEP") ? std::min(std::max(0, atoi(getenv("XSW_TAIL_SWEEP"))), 30000) : XSW_TAIL_SWEEP;
        B.tail_max = (band2 && c->T.tail_min) ? tail_max_env : 0;  // (the tail rows are k_invert_band2's to sweep)
        // strip masks: what the consumers walk when a list overflows (only the marked pixels instead of the whole raster)
        static const bool masks_off = getenv("XSW_NO_STRIP_MASKS") != nullptr;  // A/B measurements and the tests of the old route
        const size_t nstrips = (size_t)(strips_per_line * A.lines);
        if (lc.masks && nstrips <= lc.mask_strips && !masks_off) {
            B.mask_g = lc.masks; B.mask_b = lc.masks + nstrips;  // side by side: one reset (0.25 B per pixel)
            if (hipMemsetAsync(lc.masks, 0, 2 * nstrips * sizeof(unsigned long long), lc.stream) != hipSuccess) return seterr(err, XSW_EHIP, "strip-mask reset failed");
        }
        if (hipMemsetAsync(lc.list, 0, 3 * sizeof(unsigned), lc.stream) != hipSuccess) return seterr(err, XSW_EHIP, "work-list reset failed");
        const unsigned list_blocks = (unsigned)std::min<long long>(nblocks, 256 * 8);  // 8 waves per SIMD
        // k_invert_band: x = XCD lane + 8 * line group, y = tile column inside the XCD's range (see the kernel)
        const long long cols_per_xcd = (strips_per_line + 7) / 8;
        const long long band_groups = (A.lines + XSW_BAND_WG_WAVES - 1) / XSW_BAND_WG_WAVES;
        if (8 * band_groups > 0x7fffffffLL || cols_per_xcd > 65535) return seterr(err, XSW_EINVAL, "raster too large for one launch");
        const dim3 band_grid((unsigned)(8 * band_groups), (unsigned)cols_per_xcd), band_block(64 * XSW_BAND_WG_WAVES);
        if (lc.timing) timing_mark(c);
        if (count_inst) {  // statistics instantiation (counts the scored candidates)
            if (mono) hipLaunchKernelGGL((k_invert_band<T, TO, false, true>), band_grid, band_block, 0, lc.stream, c->T, B);
            else hipLaunchKernelGGL((k_invert_band<T, TO, true, true>), band_grid, band_block, 0, lc.stream, c->T, B);
        } else if (band2) {
            if (mono) hipLaunchKernelGGL((k_invert_band<T, TO, false, false, 1>), band_grid, band_block, 0, lc.stream, c->T, B);
            else hipLaunchKernelGGL((k_invert_band<T, TO, true, false, 1>), band_grid, band_block, 0, lc.stream, c->T, B);
        } else if (mono) {
            hipLaunchKernelGGL((k_invert_band<T, TO, false, false>), band_grid, band_block, 0, lc.stream, c->T, B);
        } else {
            hipLaunchKernelGGL((k_invert_band<T, TO, true, false>), band_grid, band_block, 0, lc.stream, c->T, B);
        }
        if (lc.timing) timing_mark(c);
        if (band2) {
            const dim3 b2_grid((unsigned)std::min<long long>(nblocks, 256 * XSW_BAND2_WAVES));  // XSW_BAND2_WAVES waves per SIMD, 4-wave workgroups
            if (mono) hipLaunchKernelGGL((k_invert_band2<T, TO, false>), b2_grid, band_block, 0, lc.stream, c->T, B);
            else hipLaunchKernelGGL((k_invert_band2<T, TO, true>), b2_grid, band_block, 0, lc.stream, c->T, B);
        }
        if (lc.timing) timing_mark(c);
        if (blocks3) {
            const dim3 bl_grid((unsigned)std::min<long long>(nblocks, 256 * XSW_BLOCKS_WAVES));
            if (mono) hipLaunchKernelGGL((k_invert_blocks<T, TO, false>), bl_grid, dim3(256), 0, lc.stream, c->T, B);
            else hipLaunchKernelGGL((k_invert_blocks<T, TO, true>), bl_grid, dim3(256), 0, lc.stream, c->T, B);
        }
        if (lc.timing) timing_mark(c);
        if (mono) hipLaunchKernelGGL((k_invert_list<T, TO, false>), dim3(list_blocks), dim3(256), 0, lc.stream, c->T, B);
        else hipLaunchKernelGGL((k_invert_list<T, TO, true>), dim3(list_blocks), dim3(256), 0, lc.stream, c->T, B);
        if (lc.timing) timing_mark(c);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return seterr(err, XSW_EHIP, "launch failed: %s", hipGetErrorString(e));
        return XSW_OK;
    }
    if (algo == XSW_ALGO_PRUNED && mono)
        hipLaunchKernelGGL((k_invert<T, TO, 1, false>), dim3((unsigned)nblocks), dim3(256), 0, lc.stream, c->T, A);
    else if (algo == XSW_ALGO_PRUNED)
        hipLaunchKernelGGL((k_invert<T, TO, 1>), dim3((unsigned)nblocks), dim3(256), 0, lc.stream, c->T, A);
    else
        hipLaunchKernelGGL((k_invert<T, TO, 3>), dim3((unsigned)nblocks), dim3(256), 0, lc.stream, c->T, A);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return seterr(err, XSW_EHIP, "launch failed: %s", hipGetErrorString(e));
    return XSW_OK;
}


#ifndef XSW_PAIR
#error "compile with -DXSW_PAIR=0..3"
#endif
#if XSW_PAIR == 0
int xsw_launch_invert_ff(xsw_ctx *c, const KArgs &A, int algo, const LaunchCtl &lc, std::string &err) { return launch_invert<float, float>(c, A, algo, lc, err); }
#elif XSW_PAIR == 1
int xsw_launch_invert_fd(xsw_ctx *c, const KArgs &A, int algo, const LaunchCtl &lc, std::string &err) { return launch_invert<float, double>(c, A, algo, lc, err); }
#elif XSW_PAIR == 2
int xsw_launch_invert_df(xsw_ctx *c, const KArgs &A, int algo, const LaunchCtl &lc, std::string &err) { return launch_invert<double, float>(c, A, algo, lc, err); }
#else
int xsw_launch_invert_dd(xsw_ctx *c, const KArgs &A, int algo, const LaunchCtl &lc, std::string &err) { return launch_invert<double, double>(c, A, algo, lc, err); }
#endif
