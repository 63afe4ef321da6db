// Host-side types shared by the translation units of libxsw (xsw.hip: context, LUT install, C ABI; xsw_invert_tu.hip: the
// kernel launches of one (input dtype, output dtype) pair each, compiled side by side).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "xsw.h"
#include "xsw_device.hpp"

#ifndef XSW_ARENA_KEEP
#define XSW_ARENA_KEEP ((size_t)24 << 30)
#endif
struct xsw_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    xsw::DevTables T{};
    std::vector<void *> co_allocs, cr_allocs;
    bool have_co = false, have_cr = false;
    unsigned long long *d_stats = nullptr;
    bool stats_on = false;
    bool stats_chain = false;               // xsw_stats_enable(ctx, 2): the production chain keeps running, its kernels count what they score
    bool timing_on = false;                 // xsw_timing_enable: HIP events around the kernels of every device-memory inversion
    std::vector<hipEvent_t> timing_events;  // quintuples (start, after k_invert_band, k_invert_band2, k_invert_blocks, k_invert_list) on the launch stream
    unsigned *d_list = nullptr;  // hand-over k_invert_band -> k_invert_list: [0] = count, [16..] = pixel indices
    size_t list_cap = 0;         // entries (context-owned, grown on demand: an eighth of the largest raster seen)
    unsigned long long *d_masks = nullptr;  // strip masks (2 x mask_strips words, after the lists in the same allocation)
    void *d_rec = nullptr;                  // list B's records (after the masks)
    size_t mask_strips = 0;
    double *d_ratio = nullptr;  // detrend ratio row (context-owned, grown on demand)
    size_t ratio_cap = 0;
    void *nesz_scratch = nullptr;  // xsw_nesz_flatten: column partials + means (context-owned, grown on demand)
    size_t nesz_cap = 0;
    // host-memory paths: worker w owns a stream, a page-locked staging buffer and a device staging buffer, all kept between calls
    struct Worker { hipStream_t s = nullptr; char *pin = nullptr; size_t pin_cap = 0; char *dev = nullptr; size_t dev_cap = 0; };
    std::vector<Worker> workers;
    int host_threads = 0;  // 0: XSW_HOST_THREADS or 12
    char *arena = nullptr;      // whole-raster device staging (xsw_nesz_flatten on host rasters; kept up to XSW_ARENA_KEEP bytes)
    size_t arena_cap = 0;
    std::vector<void *> host_allocs;  // xsw_host_alloc
    // host copies of the output-forming tables (xsw_expand_codes on host memory; the expansion of the host-memory paths)
    std::vector<double> h_sol, h_dual, h_wcr;
    std::vector<float> h_sol32;
    std::string err;
};


static inline void timing_mark(xsw_ctx *c)

{
    if (!c->timing_on) return;
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) == hipSuccess && hipEventRecord(e, c->stream) == hipSuccess) c->timing_events.push_back(e);
    else c->timing_on = false;  // never half a quintuple
}


static inline int seterr(std::string &e, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    e = buf;
    return code;
}

// Where an inversion launches: its stream and the work list that hands pixels from k_invert_band to k_invert_list (device
// rasters: the context's; host rasters: the worker's own, so that the chunks of different workers run side by side).
struct LaunchCtl {
    hipStream_t stream;
    unsigned *list;    // [0] = count, [16 ..] = entries; nullptr: one-kernel path
    size_t list_cap;   // entries
    bool timing;       // xsw_timing_enable events (context stream only)
    unsigned long long *masks = nullptr;  // strip masks (KArgs::mask_g, then mask_b), mask_strips words each; nullptr: none
    size_t mask_strips = 0;
    void *rec_b = nullptr;  // XSW_LIST_B_SHARE * list_cap records of XSW_REC_BYTES (KArgs::rec_b); nullptr: list B holds pixel indices
};


// Work lists of one launch, side by side in one allocation: [0..15] counters, then list G (k_invert_list) of LaunchCtl::list_cap
// entries, list B (k_invert_band2) of XSW_LIST_B_SHARE times as many and list C (k_invert_blocks) of XSW_LIST_C_SHARE times as
// many: on the scenes whose a-priori wind is far from the sigma0 contour HALF the pixels are k_invert_band2's (an overflowing
// list B sends the rest through the strip mask, where stage 1 is redone for them).  With list_cap = an eighth of the raster the
// lists take 4.5 B and list B's records 24 B per pixel of the largest raster seen.
#ifndef XSW_LIST_B_SHARE
#define XSW_LIST_B_SHARE 4
#endif
#ifndef XSW_LIST_C_SHARE
#define XSW_LIST_C_SHARE 4
#endif
#define XSW_LISTS_TOTAL (1 + XSW_LIST_B_SHARE + XSW_LIST_C_SHARE)
#define XSW_REC_BYTES 48  // sizeof(BandRec) (xsw_band.hpp; static_assert in xsw_invert_tu.hip): list B's records follow the strip masks

// One (input dtype, output dtype) pair of the inversion launches per translation unit (xsw_invert_tu.hip, -DXSW_PAIR=0..3:
// f32->f32, f32->f64, f64->f32, f64->f64), so that the four sets of kernel instantiations compile side by side.
int xsw_launch_invert_ff(xsw_ctx *c, const xsw::KArgs &A, int algo, const LaunchCtl &lc, std::string &err);
int xsw_launch_invert_fd(xsw_ctx *c, const xsw::KArgs &A, int algo, const LaunchCtl &lc, std::string &err);
int xsw_launch_invert_df(xsw_ctx *c, const xsw::KArgs &A, int algo, const LaunchCtl &lc, std::string &err);
int xsw_launch_invert_dd(xsw_ctx *c, const xsw::KArgs &A, int algo, const LaunchCtl &lc, std::string &err);
