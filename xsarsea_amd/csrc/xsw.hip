// libxsw host side: context, LUT upload, launch logic behind the C ABI of include/xsw.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <limits>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "xsw.h"
#include "xsw_device.hpp"
#include "xsw_misc.hpp"
#include "xsw_gmf.hpp"
#include "xsw_nesz.hpp"
#include "xsw_lutbuild.hpp"

using namespace xsw;

#include "xsw_host.hpp"

static thread_local std::string g_create_err;

static int fail(xsw_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((c), e_ == hipErrorOutOfMemory ? XSW_ENOMEM : XSW_EHIP, "%s: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                       \
    } while (0)

extern "C" int xsw_version(void) { return XSW_VERSION; }

extern "C" int xsw_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char *xsw_last_error(const xsw_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

extern "C" int xsw_ctx_create(int device, xsw_ctx **out)
{
    if (!out) return fail(nullptr, XSW_EINVAL, "ctx out pointer is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(nullptr, XSW_EHIP, "no HIP device available (%s)", e == hipSuccess ? "count=0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(nullptr, XSW_EINVAL, "device %d out of range [0,%d)", device, n);
    xsw_ctx *c = new xsw_ctx;
    c->device = device;
    const int rc = [&]() -> int {
        HIPCHK(nullptr, hipSetDevice(device));
        HIPCHK(nullptr, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        c->stream = c->own_stream;
        HIPCHK(nullptr, hipMalloc((void **)&c->d_stats, 8 * sizeof(unsigned long long)));
        HIPCHK(nullptr, hipMemset(c->d_stats, 0, 8 * sizeof(unsigned long long)));
        return XSW_OK;
    }();
    if (rc != XSW_OK) {  // nothing half-built is handed out or leaked
        if (c->d_stats) (void)hipFree(c->d_stats);
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
        delete c;
        return rc;
    }
    *out = c;
    return XSW_OK;
}

static void free_all(std::vector<void *> &v)
{
    for (void *p : v) (void)hipFree(p);
    v.clear();
}

extern "C" int xsw_ctx_destroy(xsw_ctx *c)
{
    if (!c) return XSW_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_all(c->co_allocs);
    free_all(c->cr_allocs);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->d_ratio) (void)hipFree(c->d_ratio);
    if (c->d_list) (void)hipFree(c->d_list);
    if (c->nesz_scratch) (void)hipFree(c->nesz_scratch);
    for (hipEvent_t e : c->timing_events) (void)hipEventDestroy(e);
    if (c->arena) (void)hipFree(c->arena);
    for (auto &w : c->workers) {
        if (w.s) { (void)hipStreamSynchronize(w.s); (void)hipStreamDestroy(w.s); }
        if (w.pin) (void)hipHostFree(w.pin);
        if (w.dev) (void)hipFree(w.dev);
    }
    for (void *p : c->host_allocs) (void)hipHostFree(p);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return XSW_OK;
}

extern "C" int xsw_set_stream(xsw_ctx *c, void *s)
{
    if (!c) return XSW_EINVAL;
    if ((hipStream_t)s != c->stream) {
        // context-owned buffers (work list, nesz scratch, ratio row) are reused by the next call: what was queued on the old
        // stream must be through with them before anything queued on the new one touches them -- ordered on the device (an
        // event the new stream waits for), the host does not block
        HIPCHK(c, hipSetDevice(c->device));
        hipEvent_t ev = nullptr;
        HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ev, c->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)s, ev, 0);
        (void)hipEventDestroy(ev);  // released once the wait has been satisfied
        if (e != hipSuccess) return fail(c, XSW_EHIP, "stream hand-over failed: %s", hipGetErrorString(e));
    }
    c->stream = (hipStream_t)s;  // NULL is a valid handle: the device's default stream
    return XSW_OK;
}

extern "C" int xsw_use_own_stream(xsw_ctx *c)
{
    if (!c) return XSW_EINVAL;
    return xsw_set_stream(c, (void *)c->own_stream);
}

static int host_thread_count(const xsw_ctx *c);
static void worker_release(xsw_ctx::Worker &w)
{
    if (w.s) (void)hipStreamSynchronize(w.s);
    if (w.pin) (void)hipHostFree(w.pin);
    if (w.dev) (void)hipFree(w.dev);
    w.pin = w.dev = nullptr;
    w.pin_cap = w.dev_cap = 0;
}

// Staging kept between calls: each worker of the host-memory paths owns a page-locked buffer and a device buffer of one chunk
// (float32 mono: ~40 MB each; float64 dual-pol: ~110 MB each), i.e. up to threads x chunk of pinned host memory per context.
// After every host-memory call the buffers beyond XSW_STAGING_KEEP_MB (default 1536 MB per context -- the default 12 workers at float32 mono chunks hold ~1.1 GB --, counted per worker as the
// larger of its page-locked and its device buffer) are released, largest first -- a later call allocates them again (a few ms each).
static void trim_staging(xsw_ctx *c)
{
    static const size_t keep = (size_t)(getenv("XSW_STAGING_KEEP_MB") ? std::max(0LL, atoll(getenv("XSW_STAGING_KEEP_MB"))) : 1536) << 20;
    // a worker counts with the larger of its two buffers: the pinned-input paths (XSW_MEM_HOST_PINNED, the pinned detrend) hold
    // no page-locked staging at all, and the device side of a worker (inputs + codes + its work lists and records) outgrows
    // the pinned side
    auto held = [](const xsw_ctx::Worker &w) { return std::max(w.pin_cap, w.dev_cap); };
    size_t total = 0;
    for (auto &w : c->workers) total += held(w);
    while (total > keep) {
        xsw_ctx::Worker *big = nullptr;
        for (auto &w : c->workers)
            if (held(w) && (!big || held(w) > held(*big))) big = &w;
        if (!big) break;
        total -= held(*big);
        worker_release(*big);
    }
}

extern "C" int xsw_set_host_threads(xsw_ctx *c, int n)
{
    if (!c || n < 0) return XSW_EINVAL;
    c->host_threads = n > 32 ? 32 : n;
    // workers the new count no longer uses give their staging back now
    const size_t keep_workers = (size_t)host_thread_count(c);
    if (c->workers.size() > keep_workers) {
        (void)hipSetDevice(c->device);
        for (size_t k = keep_workers; k < c->workers.size(); ++k) {
            worker_release(c->workers[k]);
            if (c->workers[k].s) (void)hipStreamDestroy(c->workers[k].s);
        }
        c->workers.resize(keep_workers);
    }
    return XSW_OK;
}

extern "C" int xsw_host_alloc(xsw_ctx *c, size_t bytes, void **out)
{
    if (!c || !out) return XSW_EINVAL;
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return fail(c, XSW_ENOMEM, "hipHostMalloc(%zu) failed", bytes);
    c->host_allocs.push_back(p);
    *out = p;
    return XSW_OK;
}

extern "C" int xsw_host_free(xsw_ctx *c, void *p)
{
    if (!c) return XSW_EINVAL;
    auto it = std::find(c->host_allocs.begin(), c->host_allocs.end(), p);
    if (it == c->host_allocs.end()) return fail(c, XSW_EINVAL, "xsw_host_free: not a pointer of xsw_host_alloc on this context");
    c->host_allocs.erase(it);
    HIPCHK(c, hipHostFree(p));
    return XSW_OK;
}

extern "C" int xsw_synchronize(xsw_ctx *c)
{
    if (!c) return XSW_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return XSW_OK;
}

extern "C" int xsw_stats_enable(xsw_ctx *c, int on)
{
    if (!c) return XSW_EINVAL;
    c->stats_on = on != 0;
    c->stats_chain = on == 2;  // counters of the PRODUCTION chain (xsw_stats_read_chain), not of the statistics instantiation
    return XSW_OK;
}

extern "C" int xsw_timing_enable(xsw_ctx *c, int on)
{
    if (!c) return XSW_EINVAL;
    for (hipEvent_t e : c->timing_events) (void)hipEventDestroy(e);
    c->timing_events.clear();
    c->timing_on = on != 0;
    return XSW_OK;
}

extern "C" int xsw_timing_read(xsw_ctx *c, xsw_timing *out)
{
    if (!c || !out) return XSW_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out->launches = 0;
    out->first_kernel_ms = out->second_kernel_ms = out->band2_kernel_ms = out->blocks_kernel_ms = 0.0;
    out->last_band2_pixels = 0;
    out->last_list_pixels = 0;
    out->last_blocks_pixels = 0;
    if (c->d_list) {
        unsigned cnt[3] = {0, 0, 0};
        HIPCHK(c, hipMemcpy(cnt, c->d_list, sizeof cnt, hipMemcpyDeviceToHost));
        out->last_list_pixels = (int64_t)cnt[0];
        out->last_band2_pixels = (int64_t)cnt[1];
        out->last_blocks_pixels = (int64_t)cnt[2];
    }
    for (size_t k = 0; k + 5 <= c->timing_events.size(); k += 5) {  // start, after k_invert_band, k_invert_band2, k_invert_blocks, k_invert_list
        float a = 0.f, b = 0.f, bl = 0.f, d = 0.f;
        HIPCHK(c, hipEventElapsedTime(&a, c->timing_events[k], c->timing_events[k + 1]));
        HIPCHK(c, hipEventElapsedTime(&b, c->timing_events[k + 1], c->timing_events[k + 2]));
        HIPCHK(c, hipEventElapsedTime(&bl, c->timing_events[k + 2], c->timing_events[k + 3]));
        HIPCHK(c, hipEventElapsedTime(&d, c->timing_events[k + 3], c->timing_events[k + 4]));
        out->first_kernel_ms += a;
        out->band2_kernel_ms += b;
        out->blocks_kernel_ms += bl;
        out->second_kernel_ms += d;
        out->launches += 1;
    }
    for (hipEvent_t e : c->timing_events) (void)hipEventDestroy(e);
    c->timing_events.clear();
    return XSW_OK;
}

extern "C" int xsw_stats_read(xsw_ctx *c, xsw_stats *out)
{
    if (!c || !out) return XSW_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned long long h[4];
    HIPCHK(c, hipMemcpy(h, c->d_stats, sizeof h, hipMemcpyDeviceToHost));
    out->pixels_co = h[0];
    out->cand_co = h[1];
    out->pixels_exact = h[2];
    out->pixels_cr = h[3];
    return XSW_OK;
}

extern "C" int xsw_stats_read_chain(xsw_ctx *c, xsw_chain_stats *out)
{
    if (!c || !out) return XSW_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    unsigned long long h[8];
    HIPCHK(c, hipMemcpy(h, c->d_stats, sizeof h, hipMemcpyDeviceToHost));
    out->cand_band2 = h[4];
    out->cand_blocks = h[5];
    out->cand_list = h[6];
    out->pixels_refined = h[7];
    return XSW_OK;
}

// ---------------------------------------------------------------------------------------- LUT upload
static bool strictly_ascending(const double *a, int n)
{
    for (int i = 1; i < n; ++i)
        if (!(a[i] > a[i - 1])) return false;
    return true;
}
// "uniform" to the error budget the pruned kernels' screening assumes: they score with w_i = w0 + i*step (forward
// differences) and re-score only candidates within 1e-9 (1 + |J_min| + m2) of the screening minimum with the real axis
// values, so an axis point may be off its grid position by no more than ~1e-12 relative (dJ/dw is O(10..100)): np.linspace
// axes are within a few ulps and pass; an axis stored in float32, or perturbed by 1e-7 of a step, takes the exact kernel.
static bool uniform_axis(const double *a, int n)
{
    if (n < 2) return false;
    const double step = (a[n - 1] - a[0]) / (n - 1);
    if (!(step > 0) || !std::isfinite(step)) return false;
    const double tol = 1e-12 * std::max(std::max(std::fabs(a[0]), std::fabs(a[n - 1])), step);
    for (int i = 0; i < n; ++i)
        if (!(std::fabs(a[i] - (a[0] + i * step)) <= tol)) return false;
    return true;
}
static bool all_finite(const double *a, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (!std::isfinite(a[i])) return false;
    return true;
}

template <typename V>
static int upload(xsw_ctx *c, std::vector<void *> &owner, const V *host, size_t count, const V **dev)
{
    void *p = nullptr;
    HIPCHK(c, hipMalloc(&p, count * sizeof(V) + 64));
    owner.push_back(p);
    HIPCHK(c, hipMemcpyAsync(p, host, count * sizeof(V), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *dev = (const V *)p;
    return XSW_OK;
}

// Installs a co-pol LUT.  The dense dB table [n_inc][n_wspd][n_phi] comes from the host (l->db) or is already on the
// device (d_dense, built by xsw_lut_build); the padded float64 copy, the float32 copy, the finiteness flag and max |dB|
// are produced on the device either way (k_pad_co).
static int install_co(xsw_ctx *c, const xsw_lut *l, const double *d_dense)
{
    if ((!l->db && !d_dense) || !l->inc || !l->wspd || !l->phi || l->n_inc < 1 || l->n_wspd < 1 || l->n_phi < 1)
        return fail(c, XSW_EINVAL, "co-pol LUT: null pointer or empty axis");
    if (!strictly_ascending(l->inc, l->n_inc) || !strictly_ascending(l->wspd, l->n_wspd) ||
        !strictly_ascending(l->phi, l->n_phi))
        return fail(c, XSW_EINVAL, "co-pol LUT: axes must be strictly ascending");
    if ((int64_t)l->n_wspd * l->n_phi >= (int64_t)1 << 30) return fail(c, XSW_EINVAL, "co-pol LUT too large");
    free_all(c->co_allocs);
    c->have_co = false;
    DevTables &T = c->T;
    const int nI = l->n_inc, nW = l->n_wspd, nP = l->n_phi;
    const int ppad = (nP + 3) & ~3, wpad = (nW + 3) & ~3;
    int rc;
    const size_t n_dense = (size_t)nI * nW * nP;
    const size_t n_pad = (size_t)nI * nW * ppad + (size_t)260 * ppad;  // + slack rows: kernels read 4 row groups ahead unmasked
    void *tmp_dense = nullptr;
    if (!d_dense) {
        HIPCHK(c, hipMalloc(&tmp_dense, n_dense * sizeof(double)));
        hipError_t e = hipMemcpyAsync(tmp_dense, l->db, n_dense * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) { (void)hipFree(tmp_dense); return fail(c, XSW_EHIP, "LUT upload failed: %s", hipGetErrorString(e)); }
        d_dense = (const double *)tmp_dense;
    }
    bool lut_finite = false;
    {
        double *d_co = nullptr;
        float *d_co32 = nullptr;
        unsigned long long *d_flags = nullptr, h_flags[2] = {0, 0};
        hipError_t e = hipMalloc((void **)&d_co, n_pad * sizeof(double) + 64);
        if (e == hipSuccess) { c->co_allocs.push_back(d_co); e = hipMalloc((void **)&d_co32, n_pad * sizeof(float) + 64); }
        if (e == hipSuccess) { c->co_allocs.push_back(d_co32); e = hipMalloc((void **)&d_flags, 2 * sizeof(unsigned long long)); }
        if (e == hipSuccess) e = hipMemsetAsync(d_flags, 0, 2 * sizeof(unsigned long long), c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_co + (size_t)nI * nW * ppad, 0, (size_t)260 * ppad * sizeof(double), c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_co32 + (size_t)nI * nW * ppad, 0, (size_t)260 * ppad * sizeof(float), c->stream);
        if (e == hipSuccess) {
            const long long rows = (long long)nI * nW;
            hipLaunchKernelGGL(k_pad_co, dim3((unsigned)std::min<long long>((rows + 3) / 4, 256 * 32)), dim3(256), 0, c->stream, d_dense, d_co,
                               d_co32, nP, ppad, rows, d_flags);
            e = hipGetLastError();
        }
        int *d_mono = nullptr;
        if (e == hipSuccess) e = hipMalloc((void **)&d_mono, (size_t)nI * sizeof(int) + 64);
        if (e == hipSuccess) {
            c->co_allocs.push_back(d_mono);
            std::vector<int> init((size_t)nI, nW);
            e = hipMemcpy(d_mono, init.data(), (size_t)nI * sizeof(int), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_mono_rows, dim3((unsigned)(((long long)nI * nP + 255) / 256)), dim3(256), 0, c->stream, d_dense, nI, nW, nP, d_mono);
            e = hipGetLastError();
        }
        T.mono_rows = d_mono;
        T.tail_min = nullptr;
        static const bool tail_off = getenv("XSW_NO_TAIL_CUT") != nullptr;  // A/B measurements only
        if (e == hipSuccess && !tail_off) {
            double *d_tail = nullptr;
            if (hipMalloc((void **)&d_tail, (size_t)nI * (XSW_TAIL_LEVELS + 1) * ppad * sizeof(double) + 64) == hipSuccess) {
                c->co_allocs.push_back(d_tail);
                hipLaunchKernelGGL(k_tail_min, dim3((unsigned)nI), dim3(256), 0, c->stream, d_dense, nW, nP, (int)ppad, d_mono, d_tail);
                if (hipGetLastError() == hipSuccess) T.tail_min = d_tail;
            } else (void)hipGetLastError();
        }
        // inverse of the monotone rows (co_band_pass starts its sweep from a table look-up instead of a bisection)
        T.inv_rows = nullptr; T.inv_grid = nullptr;
        const size_t inv_n = (size_t)nI * XSW_INV_BINS * ppad;
        if (e == hipSuccess && nW < 65536 && nI < 65536 && inv_n * sizeof(unsigned short) < ((size_t)1 << 32)) {
            unsigned short *d_inv = nullptr;
            double *d_grid = nullptr;
            e = hipMalloc((void **)&d_inv, inv_n * sizeof(unsigned short) + 64);
            if (e == hipSuccess) { c->co_allocs.push_back(d_inv); e = hipMalloc((void **)&d_grid, (size_t)3 * nI * sizeof(double) + 64); }
            if (e == hipSuccess) { c->co_allocs.push_back(d_grid); e = hipMemsetAsync(d_inv, 0, inv_n * sizeof(unsigned short), c->stream); }
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_inv_range, dim3((unsigned)nI), dim3(256), 0, c->stream, d_dense, nW, nP, d_mono, d_grid);
                hipLaunchKernelGGL(k_inv_rows, dim3((unsigned)(((long long)nI * nP + 255) / 256)), dim3(256), 0, c->stream, d_dense, nI, nW, nP,
                                   ppad, d_mono, d_grid, d_inv);
                e = hipGetLastError();
            }
            if (e == hipSuccess) { T.inv_rows = d_inv; T.inv_grid = d_grid; }
        }
        // block pyramid of the general kernel (co_block_search): min / max per block of XSW_BLK_R x XSW_BLK_C candidates and per
        // band of blk_g block rows (6 MB at the default size).  Absent (allocation failure, XSW_NO_BLOCKS=1: A/B measurements and
        // the tests of the old routes): the general kernel sweeps windows and falls back to the exact scan as before.
        T.blk = nullptr; T.bandmm = nullptr; T.blk4 = nullptr; T.cellmm = nullptr;
        T.nbr = (nW + XSW_BLK_R - 1) / XSW_BLK_R; T.nbc = (nP + XSW_BLK_C - 1) / XSW_BLK_C; T.nbc4 = (nP + XSW_BLK_C4 - 1) / XSW_BLK_C4;
        T.ncr = (T.nbr + XSW_CELL_R - 1) / XSW_CELL_R; T.ncc = (T.nbc + XSW_CELL_C - 1) / XSW_CELL_C;
        T.blk_g = std::max(1, 64 / T.nbc);
        T.nbands = (T.nbr + T.blk_g - 1) / T.blk_g;
        static const bool blocks_off = getenv("XSW_NO_BLOCKS") != nullptr;
        if (e == hipSuccess && !blocks_off && (long long)nI * T.nbr * T.nbc < (1LL << 31)) {
            float2 *d_blk = nullptr, *d_band = nullptr;
            const long long nblk = (long long)nI * T.nbr * T.nbc, nband = (long long)nI * T.nbands;
            hipError_t e2 = hipMalloc((void **)&d_blk, (size_t)nblk * sizeof(float2) + 64);
            if (e2 == hipSuccess) { c->co_allocs.push_back(d_blk); e2 = hipMalloc((void **)&d_band, (size_t)nband * sizeof(float2) + 64); }
            if (e2 == hipSuccess) {
                c->co_allocs.push_back(d_band);
                hipLaunchKernelGGL(k_block_minmax, dim3((unsigned)((nblk + 255) / 256)), dim3(256), 0, c->stream, d_dense, nI, nW, nP, T.nbr, T.nbc, d_blk);
                hipLaunchKernelGGL(k_band_minmax, dim3((unsigned)((nband + 255) / 256)), dim3(256), 0, c->stream, d_blk, nI, T.nbr, T.nbc, T.blk_g, T.nbands, d_band);
                e2 = hipGetLastError();
            }
            // level 1 of k_invert_blocks: cells of XSW_CELL_R x XSW_CELL_C blocks (a band over ALL directions has a sigma0 range that
            // holds nearly any s and no sector bound: 401 blocks left to bound per outlier pixel where these cells leave 70)
            float2 *d_cell = nullptr;
            const long long ncell = (long long)nI * T.ncr * T.ncc;
            if (e2 == hipSuccess) e2 = hipMalloc((void **)&d_cell, (size_t)ncell * sizeof(float2) + 64);
            if (e2 == hipSuccess) {
                c->co_allocs.push_back(d_cell);
                hipLaunchKernelGGL(k_block_minmax, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, c->stream, d_dense, nI, nW, nP, T.ncr, T.ncc, d_cell,
                                   XSW_CELL_C * XSW_BLK_C, XSW_CELL_R * XSW_BLK_R);
                e2 = hipGetLastError();
            }
            if (e2 == hipSuccess) { T.blk = d_blk; T.bandmm = d_band; T.cellmm = d_cell; }
            else (void)hipGetLastError();
            // the same per sub-block of XSW_BLK_C4 directions (k_invert_blocks bounds the quarters of a kept block before sweeping:
            // sigma0 varies faster with the direction than with the speed where the GMF saturates, so a block 16 directions wide
            // nearly always straddles the contour); 23 MB at the default size.  XSW_NO_BLK4=1: not built (A/B, tests of the old sweep)
            static const bool blk4_off = getenv("XSW_NO_BLK4") != nullptr;
            if (T.blk && !blk4_off && (long long)nI * T.nbr * T.nbc4 < (1LL << 31)) {
                float2 *d_blk4 = nullptr;
                const long long nblk4 = (long long)nI * T.nbr * T.nbc4;
                hipError_t e3 = hipMalloc((void **)&d_blk4, (size_t)nblk4 * sizeof(float2) + 64);
                if (e3 == hipSuccess) {
                    c->co_allocs.push_back(d_blk4);
                    hipLaunchKernelGGL(k_block_minmax, dim3((unsigned)((nblk4 + 255) / 256)), dim3(256), 0, c->stream, d_dense, nI, nW, nP, T.nbr, T.nbc4, d_blk4, XSW_BLK_C4);
                    e3 = hipGetLastError();
                }
                if (e3 == hipSuccess) T.blk4 = d_blk4;
                else (void)hipGetLastError();
            }
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h_flags, d_flags, sizeof h_flags, hipMemcpyDeviceToHost, c->stream);
        hipError_t se = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = se;
        if (d_flags) (void)hipFree(d_flags);
        if (tmp_dense) (void)hipFree(tmp_dense);
        if (e != hipSuccess) return fail(c, e == hipErrorOutOfMemory ? XSW_ENOMEM : XSW_EHIP, "LUT install failed: %s", hipGetErrorString(e));
        T.co = d_co;
        T.co32 = d_co32;
        lut_finite = h_flags[0] == 0;
        memcpy(&T.co_absmax, &h_flags[1], sizeof(double));
    }
    std::vector<double> wh(nW), cp(nP), sp(nP);
    for (int i = 0; i < nW; ++i) wh[i] = 0.5 * l->wspd[i];
    bool trig_ok = true;
    for (int i = 0; i < nP; ++i) {
        const double r = l->phi[i] * (M_PI / 180.0);
        cp[i] = l->cos_phi ? l->cos_phi[i] : std::cos(r);
        sp[i] = l->sin_phi ? l->sin_phi[i] : std::sin(r);
        if (std::fabs(cp[i] - std::cos(r)) > 1e-12 || std::fabs(sp[i] - std::sin(r)) > 1e-12) trig_ok = false;
    }
    if ((rc = upload(c, c->co_allocs, l->inc, nI, &T.inc))) return rc;
    if ((rc = upload(c, c->co_allocs, l->wspd, nW, &T.w))) return rc;
    if ((rc = upload(c, c->co_allocs, wh.data(), nW, &T.wh))) return rc;
    {
        std::vector<float> wh32(nW);
        for (int i = 0; i < nW; ++i) wh32[i] = (float)wh[i];
        if ((rc = upload(c, c->co_allocs, wh32.data(), (size_t)nW, &T.wh32))) return rc;
    }
    if ((rc = upload(c, c->co_allocs, l->phi, nP, &T.phi))) return rc;
    if ((rc = upload(c, c->co_allocs, cp.data(), nP, &T.cphi))) return rc;
    if ((rc = upload(c, c->co_allocs, sp.data(), nP, &T.sphi))) return rc;
    {
        std::vector<double> cs((size_t)2 * nP);
        for (int i = 0; i < nP; ++i) { cs[2 * i] = cp[i]; cs[2 * i + 1] = sp[i]; }
        if ((rc = upload(c, c->co_allocs, cs.data(), cs.size(), &T.csphi))) return rc;
        std::vector<float> cs32(cs.begin(), cs.end());  // float32 copy: the bound arithmetic of k_invert_band2 (xsw_band2.hpp)
        if ((rc = upload(c, c->co_allocs, cs32.data(), cs32.size(), &T.csphi32))) return rc;
    }
    // output-side tables: caller's values, or the host libm's (see xsw.h)
    {
        std::vector<double> od((size_t)4 * nP), ab((size_t)nW * nP), dd((size_t)4 * nW * nP);
        for (int k = 0; k < 2; ++k)
            for (int i = 0; i < nP; ++i) {
                const double r = (k ? -l->phi[i] : l->phi[i]) * (M_PI / 180.0);
                od[((size_t)k * nP + i) * 2 + 0] = l->out_dir ? l->out_dir[((size_t)k * nP + i) * 2 + 0] : std::cos(r);
                od[((size_t)k * nP + i) * 2 + 1] = l->out_dir ? l->out_dir[((size_t)k * nP + i) * 2 + 1] : std::sin(r);
            }
        for (int iw = 0; iw < nW; ++iw)
            for (int i = 0; i < nP; ++i) {
                const double w = l->wspd[iw];
                for (int k = 0; k < 2; ++k) {
                    const double er = od[((size_t)k * nP + i) * 2], ei = od[((size_t)k * nP + i) * 2 + 1];
                    const double re = w * er, im = w * ei + 0.0 * er;
                    const size_t o = (((size_t)k * nW + iw) * nP + i) * 2;
                    if (l->dual_dir) { dd[o] = l->dual_dir[o]; dd[o + 1] = l->dual_dir[o + 1]; }
                    else { const double ph = std::atan2(im, re); dd[o] = std::cos(ph); dd[o + 1] = std::sin(ph); }
                    if (k == 0) ab[(size_t)iw * nP + i] = l->abs_co ? l->abs_co[(size_t)iw * nP + i] : std::hypot(re, im);
                }
            }
        if ((rc = upload(c, c->co_allocs, od.data(), od.size(), &T.out_dir))) return rc;
        if ((rc = upload(c, c->co_allocs, ab.data(), ab.size(), &T.abs_co))) return rc;
        if ((rc = upload(c, c->co_allocs, dd.data(), dd.size(), &T.dual_dir))) return rc;
        // the co-pol winds themselves, by the store's own operations (store_pixel: w * e.x, w * e.y + 0.0 * e.x): what a grid
        // code expands to, on the device (k_expand) and on the host (expand_host)
        std::vector<double> sol((size_t)4 * nW * nP);
        for (int k = 0; k < 2; ++k)
            for (int iw = 0; iw < nW; ++iw)
                for (int i = 0; i < nP; ++i) {
                    const double w = l->wspd[iw], ex = od[((size_t)k * nP + i) * 2], ey = od[((size_t)k * nP + i) * 2 + 1];
                    const size_t o = (((size_t)k * nW + iw) * nP + i) * 2;
                    sol[o] = w * ex;
                    sol[o + 1] = w * ey + 0.0 * ex;
                }
        if ((rc = upload(c, c->co_allocs, sol.data(), sol.size(), &T.sol))) return rc;
        c->h_sol32.resize(sol.size());
        for (size_t k = 0; k < sol.size(); ++k) c->h_sol32[k] = (float)sol[k];
        c->h_sol.swap(sol);
        c->h_dual.swap(dd);
    }
    T.n_inc = nI; T.n_w = nW; T.n_phi = nP; T.phi_pad = ppad; T.w_pad = wpad;
    T.phi_180 = (180.0 - (l->phi[nP - 1] - l->phi[0])) < 2.0 ? 1 : 0;  // windspeed.py:152-156
    T.w0 = l->wspd[0];
    T.phi0 = l->phi[0];
    T.phi_last = l->phi[nP - 1];
    T.inv_wstep = nW > 1 ? (nW - 1) / (l->wspd[nW - 1] - l->wspd[0]) : 0.0;
    T.inv_dphi = nP > 1 ? (nP - 1) / (l->phi[nP - 1] - l->phi[0]) : 0.0;
    T.wstep_half = 0.5 / T.inv_wstep;  // the kernels' (w/2)-per-row step: same IEEE quotient they used to form per wave
    T.inv_nphi = 1.0 / (double)nP;
    T.inc_uniform = uniform_axis(l->inc, nI) && nI >= 2 ? 1 : 0;
    T.inc0 = l->inc[0];
    T.inv_incstep = nI > 1 ? (nI - 1) / (l->inc[nI - 1] - l->inc[0]) : 0.0;
    T.prunable = (nW >= 2 && nP >= 2 && nW < 32768 && nP < 65536 && (int64_t)nW * ppad < ((int64_t)1 << 30) && uniform_axis(l->wspd, nW) && uniform_axis(l->phi, nP) && trig_ok &&
                  (l->phi[nP - 1] - l->phi[0]) <= 360.0 + 1e-9 && lut_finite)
                     ? 1 : 0;
    T.co_off32 = ((uint64_t)nI * nW + 260) * (uint64_t)ppad * 8u < ((uint64_t)1 << 32) ? 1 : 0;
    T.band_mul24 = ((uint64_t)nI * nW <= 0xFFFFFFu && (uint64_t)(nI + 1) * XSW_INV_BINS <= 0xFFFFFFu && (uint64_t)ppad * 8u <= 0xFFFFFFu &&
                    (uint64_t)(nI + 1) * nP <= 0xFFFFFFu && (uint64_t)wpad * 8u <= 0xFFFFFFu && (uint64_t)nI * nP * wpad * 8u < ((uint64_t)1 << 32)) ? 1 : 0;
    T.blk_span_ok = (nP > 1 && (XSW_BLK_C - 1) * (l->phi[nP - 1] - l->phi[0]) / (nP - 1) < 170.0) ? 1 : 0;
    T.cell_span_ok = (nP > 1 && (XSW_CELL_C * XSW_BLK_C - 1) * (l->phi[nP - 1] - l->phi[0]) / (nP - 1) < 170.0) ? 1 : 0;
    // transposed slices for the ray scan
    double *dT = nullptr;
    HIPCHK(c, hipMalloc((void **)&dT, (size_t)nI * nP * wpad * sizeof(double) + 512 * sizeof(double)));
    c->co_allocs.push_back(dT);
    HIPCHK(c, hipMemsetAsync(dT, 0, (size_t)nI * nP * wpad * sizeof(double) + 512 * sizeof(double), c->stream));
    dim3 grid((nP + 31) / 32, (nW + 31) / 32, nI);
    hipLaunchKernelGGL(k_transpose_slices, grid, dim3(256), 0, c->stream, T.co, dT, nW, nP, ppad, wpad);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    T.coT = dT;
    c->have_co = true;
    return XSW_OK;
}

static int upload_cr(xsw_ctx *c, const xsw_lut *l)
{
    if (!l->db || !l->inc || !l->wspd || l->n_inc < 1 || l->n_wspd < 1)
        return fail(c, XSW_EINVAL, "cross-pol LUT: null pointer or empty axis");
    if (!strictly_ascending(l->inc, l->n_inc) || !strictly_ascending(l->wspd, l->n_wspd))
        return fail(c, XSW_EINVAL, "cross-pol LUT: axes must be strictly ascending");
    free_all(c->cr_allocs);
    c->have_cr = false;
    DevTables &T = c->T;
    const int nI = l->n_inc, nW = l->n_wspd, wpad = (nW + 3) & ~3;
    std::vector<double> pad((size_t)nI * wpad, 0.0), wh(nW);
    for (int r = 0; r < nI; ++r) memcpy(&pad[(size_t)r * wpad], l->db + (size_t)r * nW, nW * sizeof(double));
    for (int i = 0; i < nW; ++i) wh[i] = 0.5 * l->wspd[i];
    int rc;
    if ((rc = upload(c, c->cr_allocs, pad.data(), pad.size(), &T.cr))) return rc;
    if ((rc = upload(c, c->cr_allocs, l->inc, nI, &T.inc_cr))) return rc;
    if ((rc = upload(c, c->cr_allocs, l->wspd, nW, &T.wcr))) return rc;
    if ((rc = upload(c, c->cr_allocs, wh.data(), nW, &T.wcrh))) return rc;
    T.n_inc_cr = nI; T.n_wcr = nW; T.wcr_pad = wpad;
    c->h_wcr.assign(l->wspd, l->wspd + nW);
    T.cr_finite = all_finite(l->db, (size_t)nI * nW) ? 1 : 0;
    bool mono = T.cr_finite && nW >= 2 && uniform_axis(l->wspd, nW);
    for (int r = 0; r < nI && mono; ++r)
        for (int k = 1; k < nW; ++k)
            if (l->db[(size_t)r * nW + k] < l->db[(size_t)r * nW + k - 1]) { mono = false; break; }
    T.cr_monotone = mono ? 1 : 0;
    T.wcr0 = l->wspd[0];
    T.inv_wcrstep = nW > 1 ? (nW - 1) / (l->wspd[nW - 1] - l->wspd[0]) : 0.0;
    T.wcrstep_half = 0.5 / T.inv_wcrstep;
    T.inv_cr = nullptr; T.inv_cr_grid = nullptr;
    if (mono && nW < 65536) {  // inverse of the monotone rows (search_cr_scan): first k with row[k] >= t0 + b * width
        std::vector<unsigned short> inv((size_t)nI * XSW_INV_BINS);
        std::vector<double> grid((size_t)3 * nI);
        for (int r = 0; r < nI; ++r) {
            const double *row = l->db + (size_t)r * nW;
            const double t0 = row[0], width = (row[nW - 1] - row[0]) / (double)XSW_INV_BINS;
            const bool ok = width > 0.0 && width < 1e300;
            grid[3 * r] = ok ? t0 : 0.0; grid[3 * r + 1] = ok ? width : 0.0; grid[3 * r + 2] = ok ? 1.0 / width : 0.0;
            for (int b = 0; b < XSW_INV_BINS; ++b)
                inv[(size_t)r * XSW_INV_BINS + b] = (unsigned short)(b == 0 || !ok ? 0 : std::lower_bound(row, row + nW, std::fma((double)b, width, t0)) - row);
        }
        if ((rc = upload(c, c->cr_allocs, inv.data(), inv.size(), &T.inv_cr))) return rc;
        if ((rc = upload(c, c->cr_allocs, grid.data(), grid.size(), &T.inv_cr_grid))) return rc;
    }
    T.inc_cr_uniform = uniform_axis(l->inc, nI) && nI >= 2 ? 1 : 0;
    T.inc_cr0 = l->inc[0];
    T.inv_inccrstep = nI > 1 ? (nI - 1) / (l->inc[nI - 1] - l->inc[0]) : 0.0;
    c->have_cr = true;
    return XSW_OK;
}

extern "C" int xsw_lut_upload(xsw_ctx *c, const xsw_lut *co, const xsw_lut *cr)
{
    if (!c) return XSW_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if (co && (rc = install_co(c, co, nullptr))) return rc;
    if (cr && (rc = upload_cr(c, cr))) return rc;
    return XSW_OK;
}

extern "C" int xsw_lut_read(xsw_ctx *c, int32_t cross, double *out_db)
{
    if (!c || !out_db) return XSW_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    const DevTables &T = c->T;
    if (cross ? !c->have_cr : !c->have_co) return fail(c, XSW_ENOLUT, "lut_read: no such LUT on this context");
    if (cross)
        HIPCHK(c, hipMemcpy2DAsync(out_db, (size_t)T.n_wcr * 8, T.cr, (size_t)T.wcr_pad * 8, (size_t)T.n_wcr * 8, (size_t)T.n_inc_cr,
                                   hipMemcpyDeviceToHost, c->stream));
    else
        HIPCHK(c, hipMemcpy2DAsync(out_db, (size_t)T.n_phi * 8, T.co, (size_t)T.phi_pad * 8, (size_t)T.n_phi * 8,
                                   (size_t)T.n_inc * T.n_w, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return XSW_OK;
}

// ---------------------------------------------------------------------------------------- invert
static int dispatch_invert(xsw_ctx *c, const KArgs &A, int dtype, int out_dtype, int algo, const LaunchCtl &lc, std::string &err)
{
    if (dtype == XSW_F32 && out_dtype == XSW_F32) return xsw_launch_invert_ff(c, A, algo, lc, err);
    if (dtype == XSW_F32 && out_dtype == XSW_F64) return xsw_launch_invert_fd(c, A, algo, lc, err);
    if (dtype == XSW_F64 && out_dtype == XSW_F32) return xsw_launch_invert_df(c, A, algo, lc, err);
    return xsw_launch_invert_dd(c, A, algo, lc, err);
}

// Work list of the device-raster path: an eighth of the raster's pixels (the benchmark scene leaves 0.06 %; a scene that
// leaves more than an eighth overflows it, see k_invert_list).  A failed allocation selects the one-kernel path.
static size_t list_entries_for(long long n)
{
    static const char *test_cap = getenv("XSW_LIST_CAP_TEST");  // tests: a tiny capacity, so that the overflow route runs
    if (test_cap) return (size_t)std::max(atoll(test_cap), 16LL);
    return (size_t)std::max<long long>(n / 8, 1 << 16);
}
// strips of 64 samples a raster of n pixels in `lines` lines can have, however it is cut (as given, or into lines of 4096)
static size_t strips_for(long long n, long long lines) { return (size_t)(n / 64 + std::max<long long>(lines, n / 4096) + 64); }
static void ensure_list(xsw_ctx *c, long long n, long long lines)
{
    const size_t want = (list_entries_for(n) + 1) & ~(size_t)1, want_strips = strips_for(n, lines);  // even: the masks stay 8-byte aligned
    if (want <= c->list_cap && want_strips <= c->mask_strips) return;
    (void)hipStreamSynchronize(c->stream);  // the old list may still be in use
    if (c->d_list) (void)hipFree(c->d_list);
    c->d_list = nullptr;
    c->d_masks = nullptr;
    c->d_rec = nullptr;
    c->list_cap = c->mask_strips = 0;
    static const bool no_list = getenv("XSW_FAIL_LIST_ALLOC") != nullptr;  // tests: the allocation-failure route
    // list G (`want` entries), B (XSW_LIST_B_SHARE x want) and C (XSW_LIST_C_SHARE x want), the two strip masks (0.25 B per pixel), list B's records
    if (!no_list && hipMalloc((void **)&c->d_list, (XSW_LISTS_TOTAL * want + 16) * sizeof(unsigned) + 2 * want_strips * sizeof(unsigned long long) + (size_t)XSW_LIST_B_SHARE * want * XSW_REC_BYTES) == hipSuccess) {
        c->list_cap = want;
        c->d_masks = (unsigned long long *)(c->d_list + 16 + XSW_LISTS_TOTAL * want);
        c->d_rec = (void *)(c->d_masks + 2 * want_strips);
        c->mask_strips = want_strips;
    } else { c->d_list = nullptr; (void)hipGetLastError(); }
}

// ---- grid codes -> complex winds (xsw.h: xsw_expand_codes)
template <typename TO>
__global__ __launch_bounds__(256) void k_expand(const double *__restrict__ sol, const double *__restrict__ dual_dir, const double *__restrict__ wcr,
                                                long long plane, int n_wcr, long long n, const unsigned *__restrict__ cc, const unsigned *__restrict__ cr,
                                                typename Cx<TO>::type *__restrict__ out_co, typename Cx<TO>::type *__restrict__ out_cr)
{
    typedef typename Cx<TO>::type cx_t;
    const double nan = __builtin_nan("");
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned a = cc ? cc[i] : K_CODE_NAN;
        double co_re = nan, co_im = nan;
        bool have_co = false;
        long long k = 0;
        if (a == K_CODE_NAN_RE) co_im = 0.0;
        else if (!(a & 0x80000000u) && (long long)(a & 0x3FFFFFFFu) < plane) {  // anything else (XSW_CODE_NAN, or not a code of this LUT): (nan, nan)
            k = (long long)(a & 0x3FFFFFFFu) + (long long)((a >> 30) & 1u) * plane;
            const double2 z = ((const double2 *)sol)[k];
            co_re = z.x; co_im = z.y;
            have_co = true;
        }
        if (out_co) { cx_t z; z.x = (TO)co_re; z.y = (TO)co_im; out_co[i] = z; }
        if (out_cr && cr) {
            const unsigned b = cr[i];
            double re = nan, im = nan;
            if (b == K_CODE_NAN_RE) im = 0.0;
            else if (b & 0x80000000u) { }
            else if (b & K_CODE_PICK_CO) { re = co_re; im = co_im; }
            else if ((int)(b & K_CODE_NO_INDEX) < n_wcr) {
                const double wd = wcr[b & K_CODE_NO_INDEX];
                if (have_co) { const double2 u = ((const double2 *)dual_dir)[k]; re = wd * u.x; im = wd * u.y + 0.0 * u.x; }
                else { re = wd; im = 0.0; }
            }
            cx_t z; z.x = (TO)re; z.y = (TO)im; out_cr[i] = z;
        }
    }
}

template <typename V>
static inline void stream_store(V *p, V v)  // non-temporal when the address allows it: the output rasters are written once
{
    if (((uintptr_t)p & (sizeof(V) - 1)) == 0) __builtin_nontemporal_store(v, p);
    else memcpy(p, &v, sizeof(V));
}

// Host counterpart of store_pixel / k_expand: the same table values, the same IEEE operations (this file is compiled with
// -ffp-contract=off) -- the bits the device would have stored.  idx: optional int32[n][3] as xsw_invert's out_idx.
template <typename TO>
static void expand_host(const xsw_ctx *c, size_t n, const uint32_t *cc, const uint32_t *cr, TO *out_co, TO *out_cr, int32_t *idx)
{
    typedef TO cx_t __attribute__((ext_vector_type(2)));
    const double nan = std::numeric_limits<double>::quiet_NaN();
    const size_t plane = (size_t)c->T.n_w * c->T.n_phi;
    const int nP = c->T.n_phi;
    const double *sol = c->h_sol.data(), *dual = c->h_dual.data(), *wcr = c->h_wcr.data();
    const size_t n_wcr = c->h_wcr.size();
    for (size_t i = 0; i < n; ++i) {
        const uint32_t a = cc ? cc[i] : XSW_CODE_NAN;
        double co_re = nan, co_im = nan;
        bool have_co = false;
        size_t k = 0;
        if (a == XSW_CODE_NAN_RE) co_im = 0.0;
        else if (!(a & 0x80000000u) && (size_t)(a & 0x3FFFFFFFu) < plane) {  // anything else (XSW_CODE_NAN, or not a code of this LUT): (nan, nan)
            k = (size_t)(a & 0x3FFFFFFFu) + (size_t)((a >> 30) & 1u) * plane;
            co_re = sol[2 * k]; co_im = sol[2 * k + 1];
            have_co = true;
        }
        if (out_co) { cx_t z; z.x = (TO)co_re; z.y = (TO)co_im; stream_store((cx_t *)(out_co + 2 * i), z); }
        uint32_t b = XSW_CODE_NO_INDEX;
        if (cr) {
            b = cr[i];
            if (out_cr) {
                double re = nan, im = nan;
                if (b == XSW_CODE_NAN_RE) im = 0.0;
                else if (b & 0x80000000u) { }
                else if (b & XSW_CODE_PICK_CO) { re = co_re; im = co_im; }
                else if ((size_t)(b & XSW_CODE_NO_INDEX) < n_wcr) {
                    const double wd = wcr[b & XSW_CODE_NO_INDEX];
                    if (have_co) { const double ux = dual[2 * k], uy = dual[2 * k + 1]; re = wd * ux; im = wd * uy + 0.0 * ux; }
                    else { re = wd; im = 0.0; }
                }
                cx_t z; z.x = (TO)re; z.y = (TO)im; stream_store((cx_t *)(out_cr + 2 * i), z);
            }
        }
        if (idx) {
            const int flat = (int)(a & 0x3FFFFFFFu);
            idx[3 * i + 0] = have_co ? flat / nP : -1;
            idx[3 * i + 1] = have_co ? flat % nP : -1;
            idx[3 * i + 2] = ((b & 0x80000000u) || (b & XSW_CODE_NO_INDEX) == XSW_CODE_NO_INDEX) ? -1 : (int)(b & XSW_CODE_NO_INDEX);
        }
    }
}

static int check_codes_tables(xsw_ctx *c, const uint32_t *code_co, const uint32_t *code_cr)
{
    if (code_co && !c->have_co) return fail(c, XSW_ENOLUT, "co-pol codes given but no co-pol LUT on this context");
    if (code_cr && !c->have_cr) return fail(c, XSW_ENOLUT, "cross-pol codes given but no cross-pol LUT on this context");
    return XSW_OK;
}

static int expand_codes_on(xsw_ctx *c, hipStream_t stream, int64_t n, int32_t mem, int32_t out_dtype, const uint32_t *code_co,
                           const uint32_t *code_cr, void *out_co, void *out_cr);

extern "C" int xsw_expand_codes(xsw_ctx *c, int64_t n, int32_t mem, int32_t out_dtype, const uint32_t *code_co,
                                const uint32_t *code_cr, void *out_co, void *out_cr)
{
    if (!c) return XSW_EINVAL;
    return expand_codes_on(c, c->stream, n, mem, out_dtype, code_co, code_cr, out_co, out_cr);
}

extern "C" int xsw_expand_codes_on_stream(xsw_ctx *c, void *stream, int64_t n, int32_t out_dtype, const uint32_t *code_co,
                                          const uint32_t *code_cr, void *out_co, void *out_cr)
{
    if (!c) return XSW_EINVAL;
    return expand_codes_on(c, (hipStream_t)stream, n, XSW_MEM_DEVICE, out_dtype, code_co, code_cr, out_co, out_cr);
}

static int expand_codes_on(xsw_ctx *c, hipStream_t stream, int64_t n, int32_t mem, int32_t out_dtype, const uint32_t *code_co,
                           const uint32_t *code_cr, void *out_co, void *out_cr)
{
    if (n < 0 || (!code_co && !code_cr)) return fail(c, XSW_EINVAL, "expand_codes: no codes");
    if (out_dtype != XSW_F32 && out_dtype != XSW_F64) return fail(c, XSW_EINVAL, "out_dtype must be XSW_F32 or XSW_F64");
    if ((out_co && !code_co) || (out_cr && !code_cr)) return fail(c, XSW_EINVAL, "expand_codes: an output without its codes");
    int rc = check_codes_tables(c, code_co, code_cr);
    if (rc) return rc;
    if (n == 0) return XSW_OK;
    if (mem == XSW_MEM_DEVICE) {
        HIPCHK(c, hipSetDevice(c->device));
        const unsigned blocks = (unsigned)std::min<long long>((n + 255) / 256, 256 * 16);
        const long long plane = (long long)c->T.n_w * c->T.n_phi;
        if (out_dtype == XSW_F32)
            hipLaunchKernelGGL((k_expand<float>), dim3(blocks), dim3(256), 0, stream, c->T.sol, c->T.dual_dir, c->T.wcr, plane, c->have_cr ? c->T.n_wcr : 0, (long long)n, code_co,
                               code_cr, (Cx<float>::type *)out_co, (Cx<float>::type *)out_cr);
        else
            hipLaunchKernelGGL((k_expand<double>), dim3(blocks), dim3(256), 0, stream, c->T.sol, c->T.dual_dir, c->T.wcr, plane, c->have_cr ? c->T.n_wcr : 0, (long long)n, code_co,
                               code_cr, (Cx<double>::type *)out_co, (Cx<double>::type *)out_cr);
        HIPCHK(c, hipGetLastError());
        return XSW_OK;
    }
    if (out_dtype == XSW_F32) expand_host<float>(c, (size_t)n, code_co, code_cr, (float *)out_co, (float *)out_cr, nullptr);
    else expand_host<double>(c, (size_t)n, code_co, code_cr, (double *)out_co, (double *)out_cr, nullptr);
    return XSW_OK;
}

// ---- host-memory paths: chunks through a ring of workers (thread + stream + page-locked staging + device staging each)
static int host_thread_count(const xsw_ctx *c)
{
    int n = c->host_threads;
    if (n <= 0) {
        const char *e = getenv("XSW_HOST_THREADS");
        n = e ? atoi(e) : 12;
    }
    return std::max(1, std::min(n, 32));
}

static int worker_reserve(xsw_ctx::Worker &w, size_t pin_bytes, size_t dev_bytes, std::string &err)
{
    if (!w.s && hipStreamCreateWithFlags(&w.s, hipStreamNonBlocking) != hipSuccess) return seterr(err, XSW_EHIP, "stream create failed");
    if (pin_bytes > w.pin_cap) {
        if (w.pin) (void)hipHostFree(w.pin);
        w.pin = nullptr; w.pin_cap = 0;
        if (hipHostMalloc((void **)&w.pin, pin_bytes, hipHostMallocDefault) != hipSuccess) return seterr(err, XSW_ENOMEM, "hipHostMalloc(%zu) failed", pin_bytes);
        w.pin_cap = pin_bytes;
    }
    if (dev_bytes > w.dev_cap) {
        if (w.dev) (void)hipFree(w.dev);
        w.dev = nullptr; w.dev_cap = 0;
        if (hipMalloc((void **)&w.dev, dev_bytes) != hipSuccess) return seterr(err, XSW_ENOMEM, "hipMalloc(%zu) failed", dev_bytes);
        w.dev_cap = dev_bytes;
    }
    return XSW_OK;
}

// body(k, worker, err) -> XSW_* runs chunk k start to finish (stage, upload, launch, download, synchronise its stream, write the
// caller's output).  Chunks are dealt to min(threads, nchunks) workers; no exception crosses the ABI.
template <class Body>
static int run_chunks(xsw_ctx *c, long long nchunks, Body &&body)
{
    if (nchunks <= 0) return XSW_OK;
    const int nthreads = (int)std::min<long long>(host_thread_count(c), nchunks);
    if ((int)c->workers.size() < nthreads) c->workers.resize((size_t)nthreads);
    std::atomic<long long> next{0};
    std::atomic<int> rc{XSW_OK};
    std::mutex mu;
    std::string err;
    auto loop = [&](int wi) {
        std::string e;
        int r = hipSetDevice(c->device) == hipSuccess ? XSW_OK : seterr(e, XSW_EHIP, "hipSetDevice failed in a worker thread");
        while (!r && rc.load() == XSW_OK) {
            const long long k = next.fetch_add(1);
            if (k >= nchunks) break;
            r = body(k, c->workers[(size_t)wi], e);
        }
        if (r) {
            std::lock_guard<std::mutex> lk(mu);
            if (rc.load() == XSW_OK) { rc.store(r); err = e; }
        }
    };
    std::vector<std::thread> threads;
    int started = 0;
    if (nthreads > 1) {
        try {
            for (int t = 1; t < nthreads; ++t) { threads.emplace_back(loop, t); ++started; }
        } catch (...) { /* fewer threads than asked: the ones that started (and this one) take all the chunks */ }
    }
    loop(0);
    for (auto &t : threads) t.join();
    (void)started;
    if (rc.load() != XSW_OK) return fail(c, rc.load(), "%s", err.c_str());
    return XSW_OK;
}

extern "C" int xsw_invert(xsw_ctx *c, const xsw_invert_args *a)
{
    if (!c || !a) return XSW_EINVAL;
    if (a->lines < 0 || a->samples < 0) return fail(c, XSW_EINVAL, "negative raster shape");
    if ((a->dtype != XSW_F32 && a->dtype != XSW_F64) || (a->out_dtype != XSW_F32 && a->out_dtype != XSW_F64))
        return fail(c, XSW_EINVAL, "dtype/out_dtype must be XSW_F32 or XSW_F64");
    if (a->mem != XSW_MEM_HOST && a->mem != XSW_MEM_DEVICE && a->mem != XSW_MEM_HOST_PINNED && a->mem != XSW_MEM_DEVICE_SIGMA0_HOST)
        return fail(c, XSW_EINVAL, "bad mem kind");
    if (!a->inc) return fail(c, XSW_EINVAL, "inc is NULL");
    if (!a->sigma0_co && !a->sigma0_cr) return fail(c, XSW_EINVAL, "neither sigma0_co nor sigma0_cr given");
    if (a->sigma0_co && !c->have_co) return fail(c, XSW_ENOLUT, "sigma0_co given but no co-pol LUT uploaded");
    if (a->sigma0_cr && !c->have_cr) return fail(c, XSW_ENOLUT, "sigma0_cr given but no cross-pol LUT uploaded");
    if (a->sigma0_co && !a->out_co && !a->out_code_co) return fail(c, XSW_EINVAL, "out_co is NULL");
    if (a->algo < XSW_ALGO_AUTO || a->algo > XSW_ALGO_EXHAUSTIVE_F64) return fail(c, XSW_EINVAL, "unknown algo %d", a->algo);
    const long long n = (long long)a->lines * a->samples;
    if (n == 0) return XSW_OK;
    HIPCHK(c, hipSetDevice(c->device));

    int algo = a->algo == XSW_ALGO_AUTO ? XSW_ALGO_PRUNED : a->algo;
    if ((algo == XSW_ALGO_EXHAUSTIVE || algo == XSW_ALGO_EXHAUSTIVE_F64) && !(a->sigma0_co && c->T.prunable && !a->sigma0_cr))
        return fail(c, XSW_EINVAL, "XSW_ALGO_EXHAUSTIVE handles mono co-pol on a uniform finite LUT only");

    KArgs A{};
    A.n = n;
    A.lines = a->lines;
    A.samples = a->samples;
    A.dsig_co = a->dsig_co;
    A.inv_dsig_co = 1.0 / a->dsig_co;
    A.dsig_cr_scalar = a->dsig_cr_scalar;
    A.is_db = a->sigma0_is_db;
    A.dual_select = a->dual_select;
    if (!(std::isfinite(A.inv_dsig_co) && A.inv_dsig_co != 0.0) && algo != XSW_ALGO_EXACT) algo = XSW_ALGO_EXACT;
    if (c->stats_on) {
        HIPCHK(c, hipMemsetAsync(c->d_stats, 0, 8 * sizeof(unsigned long long), c->stream));
        A.stats = c->d_stats;
        A.stats_chain = c->stats_chain ? 1 : 0;
    }

    if (a->mem == XSW_MEM_DEVICE) {
        A.inc = a->inc; A.s_co = a->sigma0_co; A.s_cr = a->sigma0_cr; A.dsig_cr = a->dsig_cr; A.anc = a->anc;
        A.out_co = a->out_co; A.out_cr = a->out_cr; A.out_idx = a->out_idx;
        A.code_co = a->out_code_co; A.code_cr = a->out_code_cr;
        if (algo == XSW_ALGO_PRUNED) ensure_list(c, n, a->lines);
        std::string err;
        const LaunchCtl lc{c->stream, c->d_list, c->list_cap, c->timing_on, c->d_masks, c->mask_strips, c->d_rec};
        int rc;
        if (a->lines < 16 && n >= (1LL << 16)) {
            // a flat raster (a long vector of pixels: 1-D inputs arrive as one line) is re-cut into lines of 4096 samples + a
            // tail, as on the host path: the core dimension is only a loop (windspeed.py:190), and a one-line raster would leave
            // three of a workgroup's four waves idle
            const size_t es = a->dtype == XSW_F32 ? 4 : 8, os = a->out_dtype == XSW_F32 ? 8 : 16;
            const long long S = 4096, Lv = n / S, tail = n - Lv * S;
            auto shift = [](const void *p, size_t bytes) -> const void * { return p ? (const char *)p + bytes : nullptr; };
            KArgs M = A;
            M.lines = Lv; M.samples = S; M.n = Lv * S;
            rc = dispatch_invert(c, M, a->dtype, a->out_dtype, algo, lc, err);
            if (!rc && tail) {
                const size_t px = (size_t)(Lv * S);
                KArgs Tl = A;
                Tl.lines = 1; Tl.samples = tail; Tl.n = tail;
                Tl.inc = shift(A.inc, px * es); Tl.s_co = shift(A.s_co, px * es); Tl.s_cr = shift(A.s_cr, px * es);
                Tl.dsig_cr = shift(A.dsig_cr, px * es); Tl.anc = shift(A.anc, px * es * 2);
                Tl.out_co = (void *)shift(A.out_co, px * os); Tl.out_cr = (void *)shift(A.out_cr, px * os);
                Tl.out_idx = (int *)shift(A.out_idx, px * 12);
                Tl.code_co = (unsigned *)shift(A.code_co, px * 4); Tl.code_cr = (unsigned *)shift(A.code_cr, px * 4);
                rc = dispatch_invert(c, Tl, a->dtype, a->out_dtype, algo, lc, err);  // same stream, same work list: in order
            }
        } else {
            rc = dispatch_invert(c, A, a->dtype, a->out_dtype, algo, lc, err);
        }
        return rc ? fail(c, rc, "%s", err.c_str()) : XSW_OK;
    }

    if (a->mem == XSW_MEM_DEVICE_SIGMA0_HOST) {
        // Device rasters, sigma0 from the host: row chunks (~4 Mpx) through the workers -- stage the chunk's sigma0 into the worker's
        // page-locked buffer (the caller's callback may fill it: numpy's own log10 on the bit-parity route), one upload per sigma0
        // raster, the kernels on the worker's stream with the caller's device pointers advanced to the chunk, results in place.
        if (a->out_idx || a->lines < 4) return fail(c, XSW_EINVAL, "XSW_MEM_DEVICE_SIGMA0_HOST: out_idx is not supported, and the raster needs 4 lines or more");
        HIPCHK(c, hipStreamSynchronize(c->stream));  // the resident rasters' producers, and the statistics reset
        const size_t es = a->dtype == XSW_F32 ? 4 : 8, os = a->out_dtype == XSW_F32 ? 8 : 16;
        const long long lines = a->lines, samples = a->samples;
        const long long target_px = std::min<long long>(4LL << 20, std::max<long long>(1LL << 16, n / 16));
        long long lpc = (std::max<long long>((target_px + samples - 1) / samples, 4) + 3) & ~3LL;
        const long long nchunks = (lines + lpc - 1) / lpc;
        const size_t max_px = (size_t)std::min<long long>(lpc, lines) * samples;
        auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t o_co = 0, o_cr = o_co + (a->sigma0_co ? pad(max_px * es) : 0), o_end = o_cr + (a->sigma0_cr ? pad(max_px * es) : 0);
        const size_t list_cap = std::max<size_t>(max_px / 8, 1 << 14) & ~(size_t)1, mask_strips = strips_for((long long)max_px, lpc);
        const size_t o_masks = o_end + pad((XSW_LISTS_TOTAL * list_cap + 16) * sizeof(unsigned)), o_rec = o_masks + 2 * mask_strips * sizeof(unsigned long long), dev_bytes = o_rec + XSW_LIST_B_SHARE * list_cap * XSW_REC_BYTES;
        const int dtype = a->dtype, out_dtype = a->out_dtype;
        auto shift = [](const void *p, size_t bytes) -> const void * { return p ? (const char *)p + bytes : nullptr; };
        const int rc_all = run_chunks(c, nchunks, [&](long long k, xsw_ctx::Worker &w, std::string &err) -> int {
            int rc = worker_reserve(w, o_end, dev_bytes, err);
            if (rc) return rc;
            const long long l0 = k * lpc, l1 = std::min(lines, l0 + lpc);
            const size_t px0 = (size_t)l0 * samples, npx = (size_t)(l1 - l0) * samples;
            hipError_t e = hipSuccess;
            auto up = [&](int which, const void *h, size_t off) {
                if (!h || e != hipSuccess || rc) return;
                int staged = 0;
                if (a->stage) {
                    staged = a->stage(a->stage_user, which, (int64_t)px0, (int64_t)npx, w.pin + off);
                    if (staged < 0) { rc = seterr(err, XSW_EINVAL, "the staging callback failed for raster %d, pixels [%zu, %zu)", which, px0, px0 + npx); return; }
                }
                if (staged <= 0) memcpy(w.pin + off, (const char *)h + px0 * es, npx * es);
                e = hipMemcpyAsync(w.dev + off, w.pin + off, npx * es, hipMemcpyHostToDevice, w.s);
            };
            up(1, a->sigma0_co, o_co); up(2, a->sigma0_cr, o_cr);
            if (rc) return rc;
            if (e != hipSuccess) return seterr(err, XSW_EHIP, "H2D copy failed: %s", hipGetErrorString(e));
            KArgs B = A;
            B.lines = l1 - l0;
            B.n = (long long)npx;
            B.inc = shift(a->inc, px0 * es);
            B.s_co = a->sigma0_co ? w.dev + o_co : nullptr;
            B.s_cr = a->sigma0_cr ? w.dev + o_cr : nullptr;
            B.dsig_cr = shift(a->dsig_cr, px0 * es);
            B.anc = shift(a->anc, px0 * es * 2);
            B.out_co = (void *)shift(a->out_co, px0 * os);
            B.out_cr = (void *)shift(a->out_cr, px0 * os);
            B.code_co = (unsigned *)shift(a->out_code_co, px0 * 4);
            B.code_cr = (unsigned *)shift(a->out_code_cr, px0 * 4);
            const LaunchCtl lc{w.s, (unsigned *)(w.dev + o_end), list_cap, false, (unsigned long long *)(w.dev + o_masks), mask_strips, (void *)(w.dev + o_rec)};
            rc = dispatch_invert(c, B, dtype, out_dtype, algo, lc, err);
            if (rc) return rc;
            e = hipStreamSynchronize(w.s);
            return e == hipSuccess ? XSW_OK : seterr(err, XSW_EHIP, "kernel execution failed: %s", hipGetErrorString(e));
        });
        trim_staging(c);
        return rc_all;
    }

    // Host rasters.  Chunks of whole 4-line tile rows (~2 Mpx) go through the workers: stage the chunk's inputs into the
    // worker's page-locked buffer (XSW_MEM_HOST_PINNED: skipped), ONE upload, the kernels on the worker's stream, ONE download
    // of the grid codes, then the codes are expanded into the caller's rasters by the worker's thread while the other
    // workers' chunks are in other phases.
    if (c->stats_on) HIPCHK(c, hipStreamSynchronize(c->stream));  // the counters were reset on the context's stream
    const bool pinned_in = a->mem == XSW_MEM_HOST_PINNED;
    const size_t es = a->dtype == XSW_F32 ? 4 : 8;
    const bool want_co = a->sigma0_co || a->out_co || a->out_code_co;
    const bool want_cr = a->out_cr || a->out_code_cr || (a->sigma0_cr && a->out_idx);
    // A flat raster (a long vector of pixels -- the core dimension is only a loop, windspeed.py:190; 1-D inputs arrive as one
    // line) is re-cut into lines of 4096 samples + a tail: pixels are independent, and whole 4-line tiles keep the workgroups
    // full (a one-line raster leaves three of a workgroup's four waves idle) and let the chunks pipeline.
    long long lines = a->lines, samples = a->samples, tail_px = 0;
    if (lines < 16 && n >= (1LL << 16)) { samples = 4096; lines = n / samples; tail_px = n - lines * samples; }
    A.samples = samples;
    const long long target_px = std::min<long long>(2LL << 20, std::max<long long>(1LL << 16, n / 16));
    long long lines_per_chunk = samples > 0 ? (target_px + samples - 1) / samples : lines;
    lines_per_chunk = (std::max<long long>(lines_per_chunk, 4) + 3) & ~3LL;  // whole 4-line tile rows
    const long long nmain = (lines + lines_per_chunk - 1) / lines_per_chunk, nchunks = nmain + (tail_px ? 1 : 0);
    const size_t max_px = (size_t)std::max<long long>(std::min<long long>(lines_per_chunk, lines) * samples, tail_px);
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    // staging layout of a chunk (the same offsets in the page-locked and the device buffer): inputs, then codes; list after
    const size_t o_inc = 0, o_co = o_inc + pad(max_px * es), o_cr = o_co + (a->sigma0_co ? pad(max_px * es) : 0),
                 o_dsig = o_cr + (a->sigma0_cr ? pad(max_px * es) : 0), o_anc = o_dsig + (a->dsig_cr ? pad(max_px * es) : 0),
                 o_cc = o_anc + (a->anc ? pad(max_px * es * 2) : 0), o_ccr = o_cc + (want_co ? pad(max_px * 4) : 0),
                 o_end = o_ccr + (want_cr ? pad(max_px * 4) : 0);
    const size_t list_cap = std::max<size_t>(max_px / 8, 1 << 14) & ~(size_t)1, mask_strips = strips_for((long long)max_px, lines_per_chunk);
    const size_t o_masks = o_end + pad((XSW_LISTS_TOTAL * list_cap + 16) * sizeof(unsigned)), o_rec = o_masks + 2 * mask_strips * sizeof(unsigned long long), dev_bytes = o_rec + XSW_LIST_B_SHARE * list_cap * XSW_REC_BYTES;  // lists G, B and C, strip masks, list B's records
    const int dtype = a->dtype, out_dtype = a->out_dtype;
    static const bool prof = getenv("XSW_HOST_PROFILE") != nullptr;  // phase times of the pipeline on stderr (experiments)
    std::atomic<long long> t_stage{0}, t_gpu{0}, t_expand{0}, t_reserve{0};
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a0, std::chrono::steady_clock::time_point b0) {
        return (long long)std::chrono::duration_cast<std::chrono::microseconds>(b0 - a0).count(); };
    const auto t_begin = now();
    const int rc_all = run_chunks(c, nchunks, [&](long long k, xsw_ctx::Worker &w, std::string &err) -> int {
        const auto t0 = now();
        int rc = worker_reserve(w, o_end, dev_bytes, err);
        if (rc) return rc;
        const auto t1 = now();
        const bool is_tail = k >= nmain;  // the last pixels of a re-cut flat raster, as one short line
        const long long l0 = is_tail ? lines : k * lines_per_chunk, l1 = is_tail ? lines + 1 : std::min(lines, l0 + lines_per_chunk);
        const size_t px0 = (size_t)l0 * samples, npx = is_tail ? (size_t)tail_px : (size_t)(l1 - l0) * samples;
        hipError_t e = hipSuccess;
        // one raster of the chunk: host -> (page-locked staging ->) device.  The caller's staging callback may fill the staging
        // area itself; page-locked caller rasters are read by the DMA engine directly
        auto up = [&](int which, const void *h, size_t off, size_t elem) {
            if (!h || e != hipSuccess || rc) return;
            const char *src = (const char *)h + px0 * elem;
            int staged = 0;
            if (a->stage) {
                staged = a->stage(a->stage_user, which, (int64_t)px0, (int64_t)npx, w.pin + off);
                if (staged < 0) { rc = seterr(err, XSW_EINVAL, "the staging callback failed for raster %d, pixels [%zu, %zu)", which, px0, px0 + npx); return; }
            }
            if (staged > 0) src = w.pin + off;
            else if (!pinned_in) { memcpy(w.pin + off, src, npx * elem); src = w.pin + off; }
            e = hipMemcpyAsync(w.dev + off, src, npx * elem, hipMemcpyHostToDevice, w.s);
        };
        up(0, a->inc, o_inc, es); up(1, a->sigma0_co, o_co, es); up(2, a->sigma0_cr, o_cr, es); up(3, a->dsig_cr, o_dsig, es); up(4, a->anc, o_anc, es * 2);
        if (rc) return rc;
        const auto t2 = now();
        if (e != hipSuccess) return seterr(err, XSW_EHIP, "H2D copy failed: %s", hipGetErrorString(e));
        KArgs B = A;
        B.lines = l1 - l0;
        B.n = (long long)npx;
        if (is_tail) B.samples = tail_px;
        B.inc = w.dev + o_inc;
        B.s_co = a->sigma0_co ? w.dev + o_co : nullptr;
        B.s_cr = a->sigma0_cr ? w.dev + o_cr : nullptr;
        B.dsig_cr = a->dsig_cr ? w.dev + o_dsig : nullptr;
        B.anc = a->anc ? w.dev + o_anc : nullptr;
        B.code_co = want_co ? (unsigned *)(w.dev + o_cc) : nullptr;
        B.code_cr = want_cr ? (unsigned *)(w.dev + o_ccr) : nullptr;
        const LaunchCtl lc{w.s, (unsigned *)(w.dev + o_end), list_cap, false, (unsigned long long *)(w.dev + o_masks), mask_strips, (void *)(w.dev + o_rec)};
        rc = dispatch_invert(c, B, dtype, out_dtype, algo, lc, err);
        if (rc) return rc;
        if (o_end > o_cc) e = hipMemcpyAsync(w.pin + o_cc, w.dev + o_cc, o_end - o_cc, hipMemcpyDeviceToHost, w.s);
        if (e == hipSuccess) e = hipStreamSynchronize(w.s);
        if (e != hipSuccess) return seterr(err, XSW_EHIP, "kernel execution failed: %s", hipGetErrorString(e));
        const auto t3 = now();
        const uint32_t *cc = want_co ? (const uint32_t *)(w.pin + o_cc) : nullptr, *ccr = want_cr ? (const uint32_t *)(w.pin + o_ccr) : nullptr;
        if (a->out_code_co && cc) memcpy(a->out_code_co + px0, cc, npx * 4);
        if (a->out_code_cr && ccr) memcpy(a->out_code_cr + px0, ccr, npx * 4);
        int32_t *idx = a->out_idx ? a->out_idx + 3 * px0 : nullptr;
        if (a->out_co || a->out_cr || idx) {
            if (out_dtype == XSW_F32)
                expand_host<float>(c, npx, cc, ccr, a->out_co ? (float *)a->out_co + 2 * px0 : nullptr, a->out_cr ? (float *)a->out_cr + 2 * px0 : nullptr, idx);
            else
                expand_host<double>(c, npx, cc, ccr, a->out_co ? (double *)a->out_co + 2 * px0 : nullptr, a->out_cr ? (double *)a->out_cr + 2 * px0 : nullptr, idx);
        }
        if (prof) { const auto t4 = now(); t_reserve += us(t0, t1); t_stage += us(t1, t2); t_gpu += us(t2, t3); t_expand += us(t3, t4); }
        return XSW_OK;
    });
    trim_staging(c);
    if (prof)
        fprintf(stderr, "[xsw host] %lld px, %lld chunks, %d threads: wall %.2f ms; summed over workers: reserve %.2f, stage %.2f, upload+kernels+download %.2f, expand %.2f ms\n",
                n, nchunks, (int)std::min<long long>(host_thread_count(c), nchunks), us(t_begin, now()) / 1e3, t_reserve / 1e3, t_stage / 1e3, t_gpu / 1e3, t_expand / 1e3);
    return rc_all;
}

// ---------------------------------------------------------------------------------------- LUT interpolation
static bool left_neighbours(const double *x_old, int n_old, const double *x_new, int n_new, std::vector<int> &lo)
{
    // scipy interp1d: searchsorted(x_old, x_new) (side='left'), clip(1, n-1), minus one; bounds_error=True
    lo.resize(n_new);
    for (int i = 0; i < n_new; ++i) {
        const double x = x_new[i];
        if (!(x >= x_old[0] && x <= x_old[n_old - 1])) return false;
        int a = 0, b = n_old;
        while (a < b) { int m = (a + b) >> 1; if (x_old[m] < x) a = m + 1; else b = m; }
        int hi = a < 1 ? 1 : (a > n_old - 1 ? n_old - 1 : a);
        lo[i] = hi - 1;
    }
    return true;
}

// Interpolation with device-resident raw table and output (d_raw -> d_out); axes are host arrays.  Asynchronous on the
// context's stream except for the small uploads; temporaries are appended to `tmp` (freed by the caller after a sync).
static int interp_device(xsw_ctx *c, const double *d_raw, const double *inc_raw, const double *wspd_raw, const double *phi_raw,
                         int32_t n_inc_raw, int32_t n_wspd_raw, int32_t n_phi_raw, const double *inc, const double *wspd,
                         const double *phi, int32_t n_inc, int32_t n_wspd, int32_t n_phi, double *d_out, std::vector<void *> &tmp)
{
    if (!inc_raw || !wspd_raw || !inc || !wspd || n_inc_raw < 2 || n_wspd_raw < 2 || n_inc < 1 || n_wspd < 1)
        return fail(c, XSW_EINVAL, "lut_interp: null pointer or axis shorter than 2");
    const bool has_phi = n_phi_raw > 0;
    if (has_phi && (!phi_raw || !phi || n_phi_raw < 2 || n_phi < 1)) return fail(c, XSW_EINVAL, "lut_interp: bad phi axis");
    if (!strictly_ascending(inc_raw, n_inc_raw) || !strictly_ascending(wspd_raw, n_wspd_raw) ||
        (has_phi && !strictly_ascending(phi_raw, n_phi_raw)))
        return fail(c, XSW_EINVAL, "lut_interp: raw axes must be strictly ascending");
    std::vector<int> loi, low, lop;
    if (!left_neighbours(inc_raw, n_inc_raw, inc, n_inc, loi) || !left_neighbours(wspd_raw, n_wspd_raw, wspd, n_wspd, low) ||
        (has_phi && !left_neighbours(phi_raw, n_phi_raw, phi, n_phi, lop)))
        return fail(c, XSW_EINVAL, "A value in x_new is outside the interpolation range.");
    InterpArgs a{};
    int rc = XSW_OK;
    auto up = [&](const void *h, size_t bytes, const void **d) {
        void *p = nullptr;
        if (rc) return;
        if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) { rc = fail(c, XSW_ENOMEM, "lut_interp: hipMalloc failed"); return; }
        tmp.push_back(p);
        if (bytes && hipMemcpy(p, h, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = fail(c, XSW_EHIP, "lut_interp: H2D failed");
        *d = p;
    };
    a.raw = d_raw;
    up(inc_raw, (size_t)n_inc_raw * 8, (const void **)&a.xi_raw);
    up(wspd_raw, (size_t)n_wspd_raw * 8, (const void **)&a.xw_raw);
    up(inc, (size_t)n_inc * 8, (const void **)&a.xi);
    up(wspd, (size_t)n_wspd * 8, (const void **)&a.xw);
    up(loi.data(), loi.size() * 4, (const void **)&a.loi);
    up(low.data(), low.size() * 4, (const void **)&a.low);
    if (has_phi) {
        up(phi_raw, (size_t)n_phi_raw * 8, (const void **)&a.xp_raw);
        up(phi, (size_t)n_phi * 8, (const void **)&a.xp);
        up(lop.data(), lop.size() * 4, (const void **)&a.lop);
    }
    a.out = d_out;
    a.ni_raw = n_inc_raw; a.nw_raw = n_wspd_raw; a.np_raw = has_phi ? n_phi_raw : 0;
    a.ni = n_inc; a.nw = n_wspd; a.np = has_phi ? n_phi : 0;
    if (!rc) {
        const size_t n_out = (size_t)n_inc * n_wspd * (has_phi ? n_phi : 1);
        long long blocks = (long long)((n_out + 255) / 256);
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(k_lut_interp, dim3((unsigned)blocks), dim3(256), 0, c->stream, a);
        if (hipGetLastError() != hipSuccess) rc = fail(c, XSW_EHIP, "lut_interp: launch failed");
    }
    return rc;
}

extern "C" int xsw_lut_interp(xsw_ctx *c, const double *raw, const double *inc_raw, const double *wspd_raw,
                              const double *phi_raw, int32_t n_inc_raw, int32_t n_wspd_raw, int32_t n_phi_raw,
                              const double *inc, const double *wspd, const double *phi, int32_t n_inc, int32_t n_wspd,
                              int32_t n_phi, double *out)
{
    if (!c) return XSW_EINVAL;
    if (!raw || !out) return fail(c, XSW_EINVAL, "lut_interp: null pointer or axis shorter than 2");
    HIPCHK(c, hipSetDevice(c->device));
    const bool has_phi = n_phi_raw > 0;
    std::vector<void *> tmp;
    const size_t n_raw = (size_t)std::max(n_inc_raw, 0) * std::max(n_wspd_raw, 0) * (has_phi ? n_phi_raw : 1);
    const size_t n_out = (size_t)std::max(n_inc, 0) * std::max(n_wspd, 0) * (has_phi ? std::max(n_phi, 0) : 1);
    void *d_raw = nullptr, *d_out = nullptr;
    int rc = XSW_OK;
    if (hipMalloc(&d_raw, n_raw * 8 + 8) != hipSuccess || hipMalloc(&d_out, n_out * 8 + 8) != hipSuccess)
        rc = fail(c, XSW_ENOMEM, "lut_interp: hipMalloc failed");
    if (d_raw) tmp.push_back(d_raw);
    if (d_out) tmp.push_back(d_out);
    if (!rc && hipMemcpyAsync(d_raw, raw, n_raw * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail(c, XSW_EHIP, "lut_interp: H2D failed");
    if (!rc) rc = interp_device(c, (const double *)d_raw, inc_raw, wspd_raw, phi_raw, n_inc_raw, n_wspd_raw, n_phi_raw, inc, wspd, phi,
                                n_inc, n_wspd, n_phi, (double *)d_out, tmp);
    if (!rc && hipMemcpyAsync(out, d_out, n_out * 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess)
        rc = fail(c, XSW_EHIP, "lut_interp: D2H failed");
    hipError_t se = hipStreamSynchronize(c->stream);
    if (!rc && se != hipSuccess) rc = fail(c, XSW_EHIP, "lut_interp: %s", hipGetErrorString(se));
    for (void *p : tmp) (void)hipFree(p);
    return rc;
}

// ---------------------------------------------------------------------------------------- device-side LUT build
static bool same_axis(const double *a, int na, const double *b, int nb)
{
    if (na != nb) return false;
    for (int i = 0; i < na; ++i)
        if (a[i] != b[i]) return false;
    return true;
}

extern "C" int xsw_lut_build(xsw_ctx *c, int32_t gmf_id, const double *inc_raw, int32_t n_inc_raw, const double *wspd_raw,
                             int32_t n_wspd_raw, const double *phi_raw, int32_t n_phi_raw, const xsw_lut *target)
{
    if (!c) return XSW_EINVAL;
    if (gmf_id < 0 || gmf_id >= GMF_COUNT) return fail(c, XSW_EINVAL, "unknown gmf_id %d", gmf_id);
    if (!target || !inc_raw || !wspd_raw || !target->inc || !target->wspd || n_inc_raw < 1 || n_wspd_raw < 1 ||
        target->n_inc < 1 || target->n_wspd < 1)
        return fail(c, XSW_EINVAL, "lut_build: null pointer or empty axis");
    const bool copol = gmf_id <= GMF_CMODIFR2;
    if (copol != (n_phi_raw > 0) || copol != (target->n_phi > 0) || (copol && (!phi_raw || !target->phi)))
        return fail(c, XSW_EINVAL, "lut_build: a co-pol GMF needs phi axes, a cross-pol GMF must not have them");
    HIPCHK(c, hipSetDevice(c->device));
    const int npr = copol ? n_phi_raw : 1, npt = copol ? target->n_phi : 1;
    const size_t n_raw = (size_t)n_inc_raw * n_wspd_raw * npr, n_out = (size_t)target->n_inc * target->n_wspd * npt;
    std::vector<void *> tmp;
    int rc = XSW_OK;
    auto dev = [&](const void *h, size_t bytes) -> void * {
        void *p = nullptr;
        if (rc) return nullptr;
        if (hipMalloc(&p, bytes + 8) != hipSuccess) { rc = fail(c, XSW_ENOMEM, "lut_build: hipMalloc(%zu) failed", bytes); return nullptr; }
        tmp.push_back(p);
        if (h && hipMemcpy(p, h, bytes, hipMemcpyHostToDevice) != hipSuccess) rc = fail(c, XSW_EHIP, "lut_build: H2D failed");
        return p;
    };
    const double *d_i = (const double *)dev(inc_raw, (size_t)n_inc_raw * 8), *d_w = (const double *)dev(wspd_raw, (size_t)n_wspd_raw * 8);
    const double *d_p = copol ? (const double *)dev(phi_raw, (size_t)n_phi_raw * 8) : nullptr;
    double *d_raw = (double *)dev(nullptr, n_raw * 8);
    if (!rc) {
        XSW_GMF_DISPATCH(gmf_id, hipLaunchKernelGGL((k_gmf_grid<M>), dim3((unsigned)std::min<size_t>((n_raw + 255) / 256, 256 * 16)), dim3(256), 0, c->stream,
                                                    (int)gmf_id, d_i, d_w, d_p, n_inc_raw, n_wspd_raw, copol ? n_phi_raw : 0, d_raw));
        if (hipGetLastError() != hipSuccess) rc = fail(c, XSW_EHIP, "lut_build: launch failed");
    }
    // resolution change only when the grids differ (Model._normalize_lut returns the raw LUT as is otherwise)
    const bool same = same_axis(inc_raw, n_inc_raw, target->inc, target->n_inc) && same_axis(wspd_raw, n_wspd_raw, target->wspd, target->n_wspd) &&
                      (!copol || same_axis(phi_raw, n_phi_raw, target->phi, target->n_phi));
    double *d_dense = d_raw;
    if (!rc && !same) {
        d_dense = (double *)dev(nullptr, n_out * 8);
        if (!rc) rc = interp_device(c, d_raw, inc_raw, wspd_raw, phi_raw, n_inc_raw, n_wspd_raw, copol ? n_phi_raw : 0, target->inc,
                                    target->wspd, target->phi, target->n_inc, target->n_wspd, copol ? target->n_phi : 0, d_dense, tmp);
    }
    if (!rc) {
        hipLaunchKernelGGL(k_to_db, dim3((unsigned)std::min<size_t>((n_out + 255) / 256, 256 * 16)), dim3(256), 0, c->stream, d_dense, (long long)n_out);
        if (hipGetLastError() != hipSuccess) rc = fail(c, XSW_EHIP, "lut_build: launch failed");
    }
    if (!rc && copol) rc = install_co(c, target, d_dense);
    if (!rc && !copol) {  // cross-pol tables are small (a few MB): the host-side checks of upload_cr are reused
        std::vector<double> h(n_out);
        hipError_t e = hipMemcpyAsync(h.data(), d_dense, n_out * 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, XSW_EHIP, "lut_build: D2H failed: %s", hipGetErrorString(e));
        if (!rc) {
            xsw_lut t = *target;
            t.db = h.data();
            rc = upload_cr(c, &t);
        }
    }
    hipError_t se = hipStreamSynchronize(c->stream);
    if (!rc && se != hipSuccess) rc = fail(c, XSW_EHIP, "lut_build: %s", hipGetErrorString(se));
    for (void *p : tmp) (void)hipFree(p);
    return rc;
}

// ---------------------------------------------------------------------------------------- forward GMF
extern "C" int xsw_gmf_eval(xsw_ctx *c, int32_t gmf_id, int64_t n, int32_t mem, const double *inc, const double *wspd,
                            const double *phi, double *out)
{
    if (!c) return XSW_EINVAL;
    if (gmf_id < 0 || gmf_id >= GMF_COUNT) return fail(c, XSW_EINVAL, "unknown gmf_id %d", gmf_id);
    if (n < 0 || !inc || !wspd || !out) return fail(c, XSW_EINVAL, "gmf_eval: bad argument");
    if (gmf_id <= GMF_CMODIFR2 && !phi) return fail(c, XSW_EINVAL, "gmf_eval: this GMF needs phi");
    if (n == 0) return XSW_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const double *d_inc = inc, *d_w = wspd, *d_phi = phi;
    double *d_out = out;
    std::vector<void *> tmp;
    hipError_t e = hipSuccess;
    if (mem == XSW_MEM_HOST) {
        auto stage = [&](const double *h, const double **d) {
            if (e != hipSuccess || !h) return;
            void *p = nullptr;
            e = hipMalloc(&p, (size_t)n * 8);
            if (e != hipSuccess) return;
            tmp.push_back(p);
            e = hipMemcpyAsync(p, h, (size_t)n * 8, hipMemcpyHostToDevice, c->stream);
            *d = (const double *)p;
        };
        stage(inc, &d_inc); stage(wspd, &d_w); stage(phi, &d_phi);
        if (e == hipSuccess) { void *p = nullptr; e = hipMalloc(&p, (size_t)n * 8); if (e == hipSuccess) { tmp.push_back(p); d_out = (double *)p; } }
    }
    if (e == hipSuccess) {
        long long blocks = (n + 255) / 256;
        if (blocks > 256 * 16) blocks = 256 * 16;
        XSW_GMF_DISPATCH(gmf_id, hipLaunchKernelGGL((k_gmf_eval<M>), dim3((unsigned)blocks), dim3(256), 0, c->stream, (int)gmf_id, (long long)n, d_inc, d_w, d_phi, d_out));
        e = hipGetLastError();
    }
    if (mem == XSW_MEM_HOST) {
        if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream);
        hipError_t se = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = se;
        for (void *p : tmp) (void)hipFree(p);
    }
    if (e != hipSuccess) return fail(c, XSW_EHIP, "gmf_eval failed: %s", hipGetErrorString(e));
    return XSW_OK;
}

// ---------------------------------------------------------------------------------------- detrend
template <typename T, typename TO>
static void launch_detrend(hipStream_t s, const void *in, const double *ratio, const double *rinv, bool fast, void *out,
                           long long lines, long long samples)
{
    const long long quads = (samples + 3) / 4;
    const unsigned gx = (unsigned)((quads + 255) / 256);
#ifndef XSW_DETREND_WG_PER_CU
#define XSW_DETREND_WG_PER_CU 16
#endif
    long long gy = (256LL * XSW_DETREND_WG_PER_CU + gx - 1) / gx;  // workgroups per CU
    if (gy > lines) gy = lines;
    if (gy > 65535) gy = 65535;
    if (gy < 1) gy = 1;
    const long long lpb = (lines + gy - 1) / gy;
    gy = (lines + lpb - 1) / lpb;
    const dim3 grid(gx, (unsigned)gy);
    if (fast)
        hipLaunchKernelGGL((k_detrend<T, TO, 1>), grid, dim3(256), 0, s, (const T *)in, ratio, rinv, (TO *)out, lines, samples, lpb);
    else
        hipLaunchKernelGGL((k_detrend<T, TO, 0>), grid, dim3(256), 0, s, (const T *)in, ratio, rinv, (TO *)out, lines, samples, lpb);
}

extern "C" int xsw_detrend(xsw_ctx *c, int64_t lines, int64_t samples, int32_t dtype, int32_t out_dtype, int32_t mem,
                           const void *sigma0, const double *ratio_row, void *out)
{
    if (!c) return XSW_EINVAL;
    if (lines < 0 || samples < 0 || !sigma0 || !ratio_row || !out) return fail(c, XSW_EINVAL, "bad detrend argument");
    const long long n = (long long)lines * samples;
    if (n == 0) return XSW_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t es = dtype == XSW_F32 ? 4 : 8, os = out_dtype == XSW_F32 ? 4 : 8;
    // the ratio row lives in a context-owned buffer (grown on demand): no allocation on the steady-state path
    if ((size_t)samples > c->ratio_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));  // a previous asynchronous call may still read the old row
        if (c->d_ratio) (void)hipFree(c->d_ratio);
        c->d_ratio = nullptr;
        c->ratio_cap = 0;
        HIPCHK(c, hipMalloc((void **)&c->d_ratio, 2 * (size_t)samples * sizeof(double) + 64));
        c->ratio_cap = (size_t)samples;
    }
    // [ratio | RN(1/ratio)]; the fused-multiply quotient is exact only for "ordinary" divisors: check them all
    std::vector<double> both(2 * (size_t)samples);
    bool fast = true;
    for (int64_t k = 0; k < samples; ++k) {
        const double r = ratio_row[k];
        both[(size_t)k] = r;
        both[(size_t)samples + k] = 1.0 / r;
        uint64_t bits;
        memcpy(&bits, &r, 8);
        const double ar = std::fabs(r);
        if (!(ar > 0x1p-500 && ar < 0x1p500) || (bits & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull) fast = false;
    }
    double *d_rinv = c->d_ratio + samples;
    hipError_t e = hipMemcpyAsync(c->d_ratio, both.data(), 2 * (size_t)samples * sizeof(double), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // `both` is a local
    auto launch = [&](const void *din, void *dout, long long nl, hipStream_t st) {
        if (dtype == XSW_F32 && out_dtype == XSW_F32) launch_detrend<float, float>(st, din, c->d_ratio, d_rinv, fast, dout, nl, samples);
        else if (dtype == XSW_F32) launch_detrend<float, double>(st, din, c->d_ratio, d_rinv, fast, dout, nl, samples);
        else if (out_dtype == XSW_F32) launch_detrend<double, float>(st, din, c->d_ratio, d_rinv, fast, dout, nl, samples);
        else launch_detrend<double, double>(st, din, c->d_ratio, d_rinv, fast, dout, nl, samples);
        return hipGetLastError();
    };
    if (mem == XSW_MEM_DEVICE) {  // device rasters: asynchronous on the context's stream
        if (e == hipSuccess) e = launch(sigma0, out, lines, c->stream);
        if (e != hipSuccess) return fail(c, XSW_EHIP, "detrend failed: %s", hipGetErrorString(e));
        return XSW_OK;
    }
    // Host rasters (synchronous): line chunks through the workers of the host-memory path (page-locked staging, one stream per
    // worker: staging copies, uploads, kernels and downloads of different chunks overlap).
    if (e != hipSuccess) return fail(c, XSW_EHIP, "detrend failed: %s", hipGetErrorString(e));
    const bool pinned = mem == XSW_MEM_HOST_PINNED;
    const long long target_px = std::min<long long>(4LL << 20, std::max<long long>(1LL << 16, n / 16));
    long long lpc = samples > 0 ? (target_px + samples - 1) / samples : lines;
    if (lpc < 1) lpc = 1;
    const long long nchunks = (lines + lpc - 1) / lpc;
    const size_t max_px = (size_t)std::min<long long>(lpc, lines) * samples;
    const size_t o_out = ((size_t)max_px * es + 255) & ~(size_t)255, total = o_out + max_px * os;
    const int rc_det = run_chunks(c, nchunks, [&](long long k, xsw_ctx::Worker &w, std::string &err) -> int {
        int rc = worker_reserve(w, pinned ? 0 : total, total, err);
        if (rc) return rc;
        const long long l0 = k * lpc, l1 = std::min((long long)lines, l0 + lpc);
        const size_t px0 = (size_t)l0 * samples, npx = (size_t)(l1 - l0) * samples;
        const char *src = (const char *)sigma0 + px0 * es;
        char *dst = (char *)out + px0 * os;
        hipError_t ee;
        if (pinned) ee = hipMemcpyAsync(w.dev, src, npx * es, hipMemcpyHostToDevice, w.s);
        else { memcpy(w.pin, src, npx * es); ee = hipMemcpyAsync(w.dev, w.pin, npx * es, hipMemcpyHostToDevice, w.s); }
        if (ee == hipSuccess) ee = launch(w.dev, w.dev + o_out, l1 - l0, w.s);
        if (ee == hipSuccess) ee = hipMemcpyAsync(pinned ? dst : w.pin + o_out, w.dev + o_out, npx * os, hipMemcpyDeviceToHost, w.s);
        if (ee == hipSuccess) ee = hipStreamSynchronize(w.s);
        if (ee != hipSuccess) return seterr(err, XSW_EHIP, "detrend failed: %s", hipGetErrorString(ee));
        if (!pinned) memcpy(dst, w.pin + o_out, npx * os);
        return XSW_OK;
    });
    trim_staging(c);
    return rc_det;
}

// ---------------------------------------------------------------------------------------- cross-pol noise flattening
template <typename T>
static hipError_t launch_nesz(hipStream_t s, const void *noise, const void *inc, void *scratch, double *out, long long lines,
                              long long samples, int nb, long long lpb)
{
    NeszPartial *part = (NeszPartial *)scratch;
    double *col = (double *)((char *)scratch + (size_t)nb * samples * sizeof(NeszPartial));
    double *x0 = col + 2 * samples;
    double *fit = x0 + 8;  // [lines][2]
    const unsigned gx = (unsigned)((samples + 255) / 256);
    hipLaunchKernelGGL((k_nesz_colsum<T, 1>), dim3(gx, (unsigned)nb), dim3(256), 0, s, (const T *)noise, (const T *)inc, part, lines, samples, lpb);
    hipLaunchKernelGGL(k_nesz_colmean, dim3(gx), dim3(256), 0, s, part, col, samples, nb);
    hipLaunchKernelGGL(k_nesz_center, dim3(1), dim3(1024), 0, s, col, x0, samples);
    hipLaunchKernelGGL((k_nesz_fit<T>), dim3((unsigned)((lines + XSW_NESZ_LINES - 1) / XSW_NESZ_LINES)), dim3(XSW_NESZ_THREADS), 0, s, (const T *)noise, col,
                       x0, fit, lines, samples);
    // the write pass: column groups x line blocks, ~16 workgroups per CU
    const long long egx = (samples + 256LL * XSW_NESZ_EV - 1) / (256LL * XSW_NESZ_EV);
    long long enb = std::max<long long>(1, std::min<long long>((256LL * 16 + egx - 1) / egx, std::min<long long>(lines, 65535)));
    const long long elpb = (lines + enb - 1) / enb;
    enb = (lines + elpb - 1) / elpb;
    hipLaunchKernelGGL((k_nesz_eval<sizeof(T) == 4>), dim3((unsigned)egx, (unsigned)enb), dim3(256), 0, s, col, fit, out, lines, samples, elpb);
    return hipGetLastError();
}

// Moves `bytes` between a host buffer and the device through the workers' page-locked staging (32 MB pieces, side by side).
static int move_through_workers(xsw_ctx *c, void *host, void *dev, size_t bytes, bool to_device, bool pinned)
{
    const size_t piece = (size_t)32 << 20;
    const long long np = (long long)((bytes + piece - 1) / piece);
    return run_chunks(c, np, [&](long long k, xsw_ctx::Worker &w, std::string &err) -> int {
        int rc = worker_reserve(w, pinned ? 0 : piece, 0, err);
        if (rc) return rc;
        const size_t off = (size_t)k * piece, nb = std::min(piece, bytes - off);
        char *h = (char *)host + off, *d = (char *)dev + off;
        hipError_t e;
        if (to_device) {
            if (pinned) e = hipMemcpyAsync(d, h, nb, hipMemcpyHostToDevice, w.s);
            else { memcpy(w.pin, h, nb); e = hipMemcpyAsync(d, w.pin, nb, hipMemcpyHostToDevice, w.s); }
            if (e == hipSuccess) e = hipStreamSynchronize(w.s);
        } else {
            e = hipMemcpyAsync(pinned ? h : w.pin, d, nb, hipMemcpyDeviceToHost, w.s);
            if (e == hipSuccess) e = hipStreamSynchronize(w.s);
            if (e == hipSuccess && !pinned) memcpy(h, w.pin, nb);
        }
        return e == hipSuccess ? XSW_OK : seterr(err, XSW_EHIP, "staged copy failed: %s", hipGetErrorString(e));
    });
}

extern "C" int xsw_nesz_flatten(xsw_ctx *c, int64_t lines, int64_t samples, int32_t dtype, int32_t mem, const void *noise,
                                const void *inc, double *out)
{
    if (!c) return XSW_EINVAL;
    if (lines < 0 || samples < 0 || !noise || !inc || !out) return fail(c, XSW_EINVAL, "bad nesz_flatten argument");
    if (dtype != XSW_F32 && dtype != XSW_F64) return fail(c, XSW_EINVAL, "dtype must be XSW_F32 or XSW_F64");
    if (mem != XSW_MEM_HOST && mem != XSW_MEM_DEVICE && mem != XSW_MEM_HOST_PINNED) return fail(c, XSW_EINVAL, "bad mem kind");
    if (lines > 0x7fffffffLL) return fail(c, XSW_EINVAL, "raster too large for one launch");
    const long long n = (long long)lines * samples;
    if (n == 0) return XSW_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t es = dtype == XSW_F32 ? 4 : 8;
    // line blocks of the column pass: enough workgroups to fill the chip (~16 per CU), at least 8 lines each
    const long long gx = (samples + 255) / 256;
    long long nb = (256LL * 16 + gx - 1) / gx;
    nb = std::max<long long>(1, std::min<long long>(std::min<long long>(nb, (lines + 7) / 8), 65535));
    const long long lpb = (lines + nb - 1) / nb;
    nb = (lines + lpb - 1) / lpb;
    // context-owned scratch (column partials, means, centring abscissa), grown on demand: no allocation on the steady-state path
    const size_t scratch_bytes = (size_t)nb * samples * sizeof(NeszPartial) + (2 * (size_t)samples + 8 + 2 * (size_t)lines) * sizeof(double);
    if (scratch_bytes > c->nesz_cap) {
        HIPCHK(c, hipStreamSynchronize(c->stream));  // a previous call may still be using the old scratch
        if (c->nesz_scratch) (void)hipFree(c->nesz_scratch);
        c->nesz_scratch = nullptr;
        c->nesz_cap = 0;
        HIPCHK(c, hipMalloc(&c->nesz_scratch, scratch_bytes));
        c->nesz_cap = scratch_bytes;
    }
    auto launch = [&](const void *dn, const void *di, double *dout) {
        return dtype == XSW_F32 ? launch_nesz<float>(c->stream, dn, di, c->nesz_scratch, dout, lines, samples, (int)nb, lpb)
                                : launch_nesz<double>(c->stream, dn, di, c->nesz_scratch, dout, lines, samples, (int)nb, lpb);
    };
    if (mem == XSW_MEM_DEVICE) {  // asynchronous on the context's stream, like xsw_invert / xsw_detrend
        const hipError_t e = launch(noise, inc, out);
        if (e != hipSuccess) return fail(c, XSW_EHIP, "nesz_flatten failed: %s", hipGetErrorString(e));
        return XSW_OK;
    }
    // host rasters: the column means need the whole raster before the per-line pass, so the rasters are uploaded whole (through
    // the workers' page-locked staging), the four kernels run, and the result comes back the same way
    const size_t in_b = ((size_t)n * es + 255) & ~(size_t)255, need = 2 * in_b + (size_t)n * 8;
    if (need > c->arena_cap) {
        if (c->arena) (void)hipFree(c->arena);
        c->arena = nullptr;
        c->arena_cap = 0;
        if (hipMalloc((void **)&c->arena, need) != hipSuccess) return fail(c, XSW_ENOMEM, "hipMalloc(%zu) failed", need);
        c->arena_cap = need;
    }
    const bool pinned = mem == XSW_MEM_HOST_PINNED;
    char *d_noise = c->arena, *d_inc = c->arena + in_b;
    double *d_out = (double *)(c->arena + 2 * in_b);
    int rc = move_through_workers(c, (void *)noise, d_noise, (size_t)n * es, true, pinned);
    if (!rc) rc = move_through_workers(c, (void *)inc, d_inc, (size_t)n * es, true, pinned);
    if (!rc) {
        hipError_t e = launch(d_noise, d_inc, d_out);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, XSW_EHIP, "nesz_flatten failed: %s", hipGetErrorString(e));
    }
    if (!rc) rc = move_through_workers(c, out, d_out, (size_t)n * 8, false, pinned);
    trim_staging(c);
    if (c->arena_cap > XSW_ARENA_KEEP) {  // do not sit on a huge staging area
        (void)hipFree(c->arena);
        c->arena = nullptr;
        c->arena_cap = 0;
    }
    return rc;
}
