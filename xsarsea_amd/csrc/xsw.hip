// libxsw host side: context, LUT upload, launch logic behind the C ABI of include/xsw.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
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
#include "xsw_band.hpp"
#include "xsw_exhaustive.hpp"
#include "xsw_gmf.hpp"
#include "xsw_nesz.hpp"
#include "xsw_lutbuild.hpp"

using namespace xsw;

#ifndef XSW_ARENA_KEEP
#define XSW_ARENA_KEEP ((size_t)24 << 30)
#endif
struct xsw_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    DevTables T{};
    std::vector<void *> co_allocs, cr_allocs;
    bool have_co = false, have_cr = false;
    unsigned long long *d_stats = nullptr;
    bool stats_on = false;
    bool timing_on = false;                 // xsw_timing_enable: HIP events around the kernels of every device-memory inversion
    std::vector<hipEvent_t> timing_events;  // triples (start, after the first kernel, end) on the launch stream
    unsigned *d_list = nullptr;  // hand-over k_invert_band -> k_invert_list: [0] = count, [16..] = pixel indices
    size_t list_cap = 0;         // (context-owned, grown on demand)
    double *d_ratio = nullptr;  // detrend ratio row (context-owned, grown on demand)
    size_t ratio_cap = 0;
    hipStream_t s_in = nullptr, s_out = nullptr;  // upload / download streams of the host-memory path (lazily created)
    char *arena = nullptr;      // device staging of the host-memory path (context-owned, grown on demand, kept between
    size_t arena_cap = 0;       // calls up to XSW_ARENA_KEEP bytes: hipMalloc/hipFree of GBs per call cost more than the copies)
    std::string err;
};

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
        HIPCHK(nullptr, hipMalloc((void **)&c->d_stats, 4 * sizeof(unsigned long long)));
        HIPCHK(nullptr, hipMemset(c->d_stats, 0, 4 * sizeof(unsigned long long)));
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
    for (hipEvent_t e : c->timing_events) (void)hipEventDestroy(e);
    if (c->arena) (void)hipFree(c->arena);
    if (c->s_in) (void)hipStreamDestroy(c->s_in);
    if (c->s_out) (void)hipStreamDestroy(c->s_out);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return XSW_OK;
}

extern "C" int xsw_set_stream(xsw_ctx *c, void *s)
{
    if (!c) return XSW_EINVAL;
    c->stream = (hipStream_t)s;  // NULL is a valid handle: the device's default stream
    return XSW_OK;
}

extern "C" int xsw_use_own_stream(xsw_ctx *c)
{
    if (!c) return XSW_EINVAL;
    c->stream = c->own_stream;
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
    out->first_kernel_ms = out->second_kernel_ms = 0.0;
    out->last_list_pixels = 0;
    if (c->d_list) {
        unsigned cnt = 0;
        HIPCHK(c, hipMemcpy(&cnt, c->d_list, sizeof cnt, hipMemcpyDeviceToHost));
        out->last_list_pixels = (int64_t)cnt;
    }
    for (size_t k = 0; k + 3 <= c->timing_events.size(); k += 3) {
        float a = 0.f, b = 0.f;
        HIPCHK(c, hipEventElapsedTime(&a, c->timing_events[k], c->timing_events[k + 1]));
        HIPCHK(c, hipEventElapsedTime(&b, c->timing_events[k + 1], c->timing_events[k + 2]));
        out->first_kernel_ms += a;
        out->second_kernel_ms += b;
        out->launches += 1;
    }
    for (hipEvent_t e : c->timing_events) (void)hipEventDestroy(e);
    c->timing_events.clear();
    return XSW_OK;
}

static void timing_mark(xsw_ctx *c)
{
    if (!c->timing_on) return;
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) == hipSuccess && hipEventRecord(e, c->stream) == hipSuccess) c->timing_events.push_back(e);
    else c->timing_on = false;  // never half a triple
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
template <typename T, typename TO>
static int launch_invert(xsw_ctx *c, const KArgs &A, int algo)
{
    // k_invert grid: 8 XCD lanes x ceil(columns/8) tile columns x line groups (see the kernel)
    const long long strips_per_line = (A.samples + 63) / 64, line_groups = (A.lines + 3) / 4;
    const long long nblocks = 8 * ((strips_per_line + 7) / 8) * line_groups;
    if (nblocks > 0x7fffffffLL) return fail(c, XSW_EINVAL, "raster too large for one launch");
    if (algo == XSW_ALGO_EXHAUSTIVE || algo == XSW_ALGO_EXHAUSTIVE_F64)
        return launch_exhaustive<T, TO>(c->T, A, c->stream, algo == XSW_ALGO_EXHAUSTIVE) == hipSuccess
                                                ? XSW_OK : fail(c, XSW_EHIP, "exhaustive launch failed: %s", hipGetErrorString(hipGetLastError()));
    // Two-kernel fast path: k_invert_band finishes every pixel the band rule decides (monotone LUT rows, finite inputs,
    // unique minimum; cross-pol by the interval rule) and appends the rest to a work list; k_invert_list inverts those.
    static const bool band_off = getenv("XSW_NO_BAND") != nullptr;  // experiments / A-B measurements only
    if (algo == XSW_ALGO_PRUNED && !band_off && A.s_co && c->T.prunable && c->T.mono_rows && c->T.inv_rows && c->T.co_off32 &&
        (!A.s_cr || c->T.cr_monotone) && A.n < (1LL << 32)) {
        if ((size_t)A.n > c->list_cap) {
            if (c->d_list) (void)hipFree(c->d_list);
            c->d_list = nullptr;
            c->list_cap = 0;
            if (hipMalloc((void **)&c->d_list, ((size_t)A.n + 16) * sizeof(unsigned)) != hipSuccess)
                return fail(c, XSW_ENOMEM, "hipMalloc(work list, %lld) failed", A.n);
            c->list_cap = (size_t)A.n;
        }
        KArgs B = A;
        B.list_count = c->d_list;
        B.list = c->d_list + 16;
        HIPCHK(c, hipMemsetAsync(c->d_list, 0, sizeof(unsigned), c->stream));
        const unsigned list_blocks = (unsigned)std::min<long long>(nblocks, 256 * 8);  // 8 waves per SIMD
        // k_invert_band: x = XCD lane + 8 * line group, y = tile column inside the XCD's range (see the kernel)
        const long long cols_per_xcd = (strips_per_line + 7) / 8;
        const long long band_groups = (A.lines + XSW_BAND_WG_WAVES - 1) / XSW_BAND_WG_WAVES;
        if (8 * band_groups > 0x7fffffffLL || cols_per_xcd > 65535) return fail(c, XSW_EINVAL, "raster too large for one launch");
        const dim3 band_grid((unsigned)(8 * band_groups), (unsigned)cols_per_xcd), band_block(64 * XSW_BAND_WG_WAVES);
        timing_mark(c);
        if (A.stats) {  // statistics instantiation (counts the scored candidates)
            if (!A.s_cr && !A.out_cr) hipLaunchKernelGGL((k_invert_band<T, TO, false, true>), band_grid, band_block, 0, c->stream, c->T, B);
            else hipLaunchKernelGGL((k_invert_band<T, TO, true, true>), band_grid, band_block, 0, c->stream, c->T, B);
            timing_mark(c);
            if (!A.s_cr && !A.out_cr) hipLaunchKernelGGL((k_invert_list<T, TO, false>), dim3(list_blocks), dim3(256), 0, c->stream, c->T, B);
            else hipLaunchKernelGGL((k_invert_list<T, TO, true>), dim3(list_blocks), dim3(256), 0, c->stream, c->T, B);
        } else if (!A.s_cr && !A.out_cr) {
            hipLaunchKernelGGL((k_invert_band<T, TO, false, false>), band_grid, band_block, 0, c->stream, c->T, B);
            timing_mark(c);
            hipLaunchKernelGGL((k_invert_list<T, TO, false>), dim3(list_blocks), dim3(256), 0, c->stream, c->T, B);
        } else {
            hipLaunchKernelGGL((k_invert_band<T, TO, true, false>), band_grid, band_block, 0, c->stream, c->T, B);
            timing_mark(c);
            hipLaunchKernelGGL((k_invert_list<T, TO, true>), dim3(list_blocks), dim3(256), 0, c->stream, c->T, B);
        }
        timing_mark(c);
        HIPCHK(c, hipGetLastError());
        return XSW_OK;
    }
    if (algo == XSW_ALGO_PRUNED && !A.s_cr && !A.out_cr)
        hipLaunchKernelGGL((k_invert<T, TO, 1, false>), dim3((unsigned)nblocks), dim3(256), 0, c->stream, c->T, A);
    else if (algo == XSW_ALGO_PRUNED)
        hipLaunchKernelGGL((k_invert<T, TO, 1>), dim3((unsigned)nblocks), dim3(256), 0, c->stream, c->T, A);
    else
        hipLaunchKernelGGL((k_invert<T, TO, 3>), dim3((unsigned)nblocks), dim3(256), 0, c->stream, c->T, A);
    HIPCHK(c, hipGetLastError());
    return XSW_OK;
}

static int dispatch_invert(xsw_ctx *c, const KArgs &A, int dtype, int out_dtype, int algo)
{
    if (dtype == XSW_F32 && out_dtype == XSW_F32) return launch_invert<float, float>(c, A, algo);
    if (dtype == XSW_F32 && out_dtype == XSW_F64) return launch_invert<float, double>(c, A, algo);
    if (dtype == XSW_F64 && out_dtype == XSW_F32) return launch_invert<double, float>(c, A, algo);
    return launch_invert<double, double>(c, A, algo);
}

extern "C" int xsw_invert(xsw_ctx *c, const xsw_invert_args *a)
{
    if (!c || !a) return XSW_EINVAL;
    if (a->lines < 0 || a->samples < 0) return fail(c, XSW_EINVAL, "negative raster shape");
    if ((a->dtype != XSW_F32 && a->dtype != XSW_F64) || (a->out_dtype != XSW_F32 && a->out_dtype != XSW_F64))
        return fail(c, XSW_EINVAL, "dtype/out_dtype must be XSW_F32 or XSW_F64");
    if (a->mem != XSW_MEM_HOST && a->mem != XSW_MEM_DEVICE) return fail(c, XSW_EINVAL, "bad mem kind");
    if (!a->inc) return fail(c, XSW_EINVAL, "inc is NULL");
    if (!a->sigma0_co && !a->sigma0_cr) return fail(c, XSW_EINVAL, "neither sigma0_co nor sigma0_cr given");
    if (a->sigma0_co && !c->have_co) return fail(c, XSW_ENOLUT, "sigma0_co given but no co-pol LUT uploaded");
    if (a->sigma0_cr && !c->have_cr) return fail(c, XSW_ENOLUT, "sigma0_cr given but no cross-pol LUT uploaded");
    if (a->sigma0_co && !a->out_co) return fail(c, XSW_EINVAL, "out_co is NULL");
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
        HIPCHK(c, hipMemsetAsync(c->d_stats, 0, 4 * sizeof(unsigned long long), c->stream));
        A.stats = c->d_stats;
    }

    if (a->mem == XSW_MEM_DEVICE) {
        A.inc = a->inc; A.s_co = a->sigma0_co; A.s_cr = a->sigma0_cr; A.dsig_cr = a->dsig_cr; A.anc = a->anc;
        A.out_co = a->out_co; A.out_cr = a->out_cr; A.out_idx = a->out_idx;
        return dispatch_invert(c, A, a->dtype, a->out_dtype, algo);
    }

    // Host rasters: device buffers for the whole raster, work cut into chunks of whole lines and software-
    // pipelined so that the GPU inverts chunk k while the host side moves chunk k+1 in and chunk k-1 out
    // (pageable hipMemcpyAsync blocks the host but not the device).  Three streams (uploads, kernels, downloads) and two
    // events per chunk.
    const size_t es = a->dtype == XSW_F32 ? 4 : 8, os = a->out_dtype == XSW_F32 ? 8 : 16;
    int rc = XSW_OK;
    // one arena for all staging rasters, 256-byte aligned slots
    size_t need = 0;
    auto slot = [&](bool want, size_t bytes) { const size_t o = need; if (want) need += (bytes + 255) & ~(size_t)255; return want ? o : (size_t)-1; };
    const size_t o_inc = slot(true, n * es), o_co = slot(a->sigma0_co != nullptr, n * es), o_cr = slot(a->sigma0_cr != nullptr, n * es),
                 o_dsig = slot(a->dsig_cr != nullptr, n * es), o_anc = slot(a->anc != nullptr, n * es * 2),
                 o_oco = slot(a->out_co != nullptr, n * os), o_ocr = slot(a->out_cr != nullptr, n * os),
                 o_idx = slot(a->out_idx != nullptr, n * 12);
    if (need > c->arena_cap) {
        if (c->arena) (void)hipFree(c->arena);
        c->arena = nullptr;
        c->arena_cap = 0;
        if (hipMalloc((void **)&c->arena, need) != hipSuccess) return fail(c, XSW_ENOMEM, "hipMalloc(%zu) failed", need);
        c->arena_cap = need;
    }
    auto at = [&](size_t o) -> void * { return o == (size_t)-1 ? nullptr : (void *)(c->arena + o); };
    void *d_inc = at(o_inc), *d_co = at(o_co), *d_cr = at(o_cr), *d_dsig = at(o_dsig), *d_anc = at(o_anc), *d_oco = at(o_oco),
         *d_ocr = at(o_ocr), *d_idx = at(o_idx);

    // ~8 Mpx per chunk for large rasters; mid-size ones are still cut in ~8 chunks (>= 0.5 Mpx) so that they pipeline too
    const long long target_px = std::min<long long>(8LL << 20, std::max<long long>(1LL << 19, (long long)(n / 8)));
    long long lines_per_chunk = a->samples > 0 ? (target_px + a->samples - 1) / a->samples : a->lines;
    if (lines_per_chunk < 4) lines_per_chunk = 4;
    lines_per_chunk = (lines_per_chunk + 3) & ~3LL;  // whole 4-line tile rows
    const long long nchunks = (a->lines + lines_per_chunk - 1) / lines_per_chunk;
    std::vector<hipEvent_t> done((size_t)nchunks, nullptr), ready((size_t)nchunks, nullptr);
    if (!rc && ((!c->s_out && hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking) != hipSuccess) ||
                (!c->s_in && hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking) != hipSuccess)))
        rc = fail(c, XSW_EHIP, "stream create failed");
    const hipStream_t s_out = c->s_out, s_in = c->s_in;
    auto h2d = [&](void *d, const void *h, size_t off, size_t bytes) {
        if (!rc && h && hipMemcpyAsync((char *)d + off, (const char *)h + off, bytes, hipMemcpyHostToDevice, s_in) != hipSuccess)
            rc = fail(c, XSW_EHIP, "H2D copy failed");
    };
    // Downloads: pageable hipMemcpyAsync blocks the calling host thread for the length of the copy, so with more than
    // two chunks they are issued by a second host thread -- uploads + launches and downloads then proceed side by side
    // (one thread doing both is host-bound as soon as the outputs are as large as the inputs, e.g. complex128).
    int drc = XSW_OK;          // downloader's status, merged after the join
    std::string derr;
    auto drain = [&](long long k) {  // outputs of chunk k -> host, once its kernel has finished
        const long long l0 = k * lines_per_chunk, l1 = std::min((long long)a->lines, l0 + lines_per_chunk);
        const size_t px0 = (size_t)l0 * a->samples, npx = (size_t)(l1 - l0) * a->samples;
        auto d2h = [&](void *h, const void *d, size_t off, size_t bytes) {
            if (!drc && h && hipMemcpyAsync((char *)h + off, (const char *)d + off, bytes, hipMemcpyDeviceToHost, s_out) != hipSuccess) {
                drc = XSW_EHIP;
                derr = "D2H copy failed";
            }
        };
        if (!drc && hipStreamWaitEvent(s_out, done[(size_t)k], 0) != hipSuccess) { drc = XSW_EHIP; derr = "stream wait failed"; }
        d2h(a->out_co, d_oco, px0 * os, npx * os);
        d2h(a->out_cr, d_ocr, px0 * os, npx * os);
        d2h(a->out_idx, d_idx, px0 * 12, npx * 12);
    };
    std::mutex mu;
    std::condition_variable cv;
    long long launched = 0;   // chunks whose kernel and `done` event are enqueued (guarded by mu)
    bool stop = false;
    bool threaded = nchunks > 2 && !rc;
    std::thread downloader;
    if (threaded) try {
        downloader = std::thread([&] {
            if (hipSetDevice(c->device) != hipSuccess) { drc = XSW_EHIP; derr = "hipSetDevice failed in the download thread"; }
            for (long long k = 0; k < nchunks; ++k) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return launched > k || stop; });
                    if (launched <= k) return;  // the launch loop gave up
                }
                drain(k);
            }
        });
    } catch (...) {  // no exception crosses the ABI: fall back to issuing the downloads from this thread
        threaded = false;
    }
    for (long long k = 0; k < nchunks && !rc; ++k) {
        const long long l0 = k * lines_per_chunk, l1 = std::min((long long)a->lines, l0 + lines_per_chunk);
        const size_t px0 = (size_t)l0 * a->samples, npx = (size_t)(l1 - l0) * a->samples;
        h2d(d_inc, a->inc, px0 * es, npx * es);
        h2d(d_co, a->sigma0_co, px0 * es, npx * es);
        h2d(d_cr, a->sigma0_cr, px0 * es, npx * es);
        h2d(d_dsig, a->dsig_cr, px0 * es, npx * es);
        h2d(d_anc, a->anc, px0 * es * 2, npx * es * 2);
        // the uploads ride their own stream (in the kernels' stream they would queue up behind the previous chunk's
        // kernel instead of overlapping it); the kernel of chunk k waits for its inputs only
        if (!rc && (hipEventCreateWithFlags(&ready[(size_t)k], hipEventDisableTiming) != hipSuccess ||
                    hipEventRecord(ready[(size_t)k], s_in) != hipSuccess ||
                    hipStreamWaitEvent(c->stream, ready[(size_t)k], 0) != hipSuccess))
            rc = fail(c, XSW_EHIP, "event record failed");
        KArgs B = A;
        B.lines = l1 - l0;
        B.n = (long long)npx;
        B.inc = (const char *)d_inc + px0 * es;
        B.s_co = d_co ? (const char *)d_co + px0 * es : nullptr;
        B.s_cr = d_cr ? (const char *)d_cr + px0 * es : nullptr;
        B.dsig_cr = d_dsig ? (const char *)d_dsig + px0 * es : nullptr;
        B.anc = d_anc ? (const char *)d_anc + px0 * es * 2 : nullptr;
        B.out_co = d_oco ? (char *)d_oco + px0 * os : nullptr;
        B.out_cr = d_ocr ? (char *)d_ocr + px0 * os : nullptr;
        B.out_idx = d_idx ? (int *)((char *)d_idx + px0 * 12) : nullptr;
        if (!rc) rc = dispatch_invert(c, B, a->dtype, a->out_dtype, algo);
        if (!rc && (hipEventCreateWithFlags(&done[(size_t)k], hipEventDisableTiming) != hipSuccess ||
                    hipEventRecord(done[(size_t)k], c->stream) != hipSuccess))
            rc = fail(c, XSW_EHIP, "event record failed");
        if (threaded) {
            if (!rc) {
                std::lock_guard<std::mutex> lk(mu);
                launched = k + 1;
            }
            cv.notify_one();
        } else if (k > 0 && !rc) drain(k - 1);  // overlaps with the kernel of chunk k
    }
    if (threaded) {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_one();
        downloader.join();
    } else if (!rc && nchunks > 0) drain(nchunks - 1);
    if (!rc && drc) rc = fail(c, drc, "%s", derr.c_str());
    hipError_t se = hipStreamSynchronize(c->stream);
    hipError_t so = s_out ? hipStreamSynchronize(s_out) : hipSuccess;
    if (!rc && (se != hipSuccess || so != hipSuccess))
        rc = fail(c, XSW_EHIP, "kernel execution failed: %s", hipGetErrorString(se != hipSuccess ? se : so));
    hipError_t si = s_in ? hipStreamSynchronize(s_in) : hipSuccess;
    if (!rc && si != hipSuccess) rc = fail(c, XSW_EHIP, "upload failed: %s", hipGetErrorString(si));
    for (hipEvent_t e : done) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ready) if (e) (void)hipEventDestroy(e);
    if (c->arena_cap > XSW_ARENA_KEEP) {  // do not sit on a huge staging area
        (void)hipFree(c->arena);
        c->arena = nullptr;
        c->arena_cap = 0;
    }
    return rc;
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
        hipLaunchKernelGGL(k_gmf_grid, dim3((unsigned)std::min<size_t>((n_raw + 255) / 256, 256 * 16)), dim3(256), 0, c->stream, (int)gmf_id,
                           d_i, d_w, d_p, n_inc_raw, n_wspd_raw, copol ? n_phi_raw : 0, d_raw);
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
        hipLaunchKernelGGL(k_gmf_eval, dim3((unsigned)blocks), dim3(256), 0, c->stream, (int)gmf_id, (long long)n, d_inc, d_w, d_phi, d_out);
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
    auto launch = [&](const void *din, void *dout, long long nl) {
        if (dtype == XSW_F32 && out_dtype == XSW_F32) launch_detrend<float, float>(c->stream, din, c->d_ratio, d_rinv, fast, dout, nl, samples);
        else if (dtype == XSW_F32) launch_detrend<float, double>(c->stream, din, c->d_ratio, d_rinv, fast, dout, nl, samples);
        else if (out_dtype == XSW_F32) launch_detrend<double, float>(c->stream, din, c->d_ratio, d_rinv, fast, dout, nl, samples);
        else launch_detrend<double, double>(c->stream, din, c->d_ratio, d_rinv, fast, dout, nl, samples);
        return hipGetLastError();
    };
    if (mem == XSW_MEM_DEVICE) {  // device rasters: asynchronous on the context's stream
        if (e == hipSuccess) e = launch(sigma0, out, lines);
        if (e != hipSuccess) return fail(c, XSW_EHIP, "detrend failed: %s", hipGetErrorString(e));
        return XSW_OK;
    }
    // Host rasters (synchronous): the same three-stream pipeline as xsw_invert -- uploads, kernels and downloads of
    // successive line chunks overlap; the downloads are issued by a second host thread (a pageable copy blocks its caller).
    if (e != hipSuccess) return fail(c, XSW_EHIP, "detrend failed: %s", hipGetErrorString(e));
    const size_t in_bytes = ((size_t)n * es + 255) & ~(size_t)255, need = in_bytes + (size_t)n * os;
    if (need > c->arena_cap) {
        if (c->arena) (void)hipFree(c->arena);
        c->arena = nullptr;
        c->arena_cap = 0;
        if (hipMalloc((void **)&c->arena, need) != hipSuccess) return fail(c, XSW_ENOMEM, "hipMalloc(%zu) failed", need);
        c->arena_cap = need;
    }
    if ((!c->s_out && hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking) != hipSuccess) ||
        (!c->s_in && hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking) != hipSuccess))
        return fail(c, XSW_EHIP, "stream create failed");
    char *t_in = c->arena, *t_out = c->arena + in_bytes;
    const long long target_px = std::min<long long>(16LL << 20, std::max<long long>(1LL << 19, n / 8));
    long long lpc = samples > 0 ? (target_px + samples - 1) / samples : lines;
    if (lpc < 1) lpc = 1;
    const long long nchunks = (lines + lpc - 1) / lpc;
    std::vector<hipEvent_t> done((size_t)nchunks, nullptr), ready((size_t)nchunks, nullptr);
    hipError_t de = hipSuccess;  // downloader's status
    auto drain = [&](long long k) {
        const long long l0 = k * lpc, l1 = std::min((long long)lines, l0 + lpc);
        const size_t px0 = (size_t)l0 * samples, npx = (size_t)(l1 - l0) * samples;
        if (de == hipSuccess) de = hipStreamWaitEvent(c->s_out, done[(size_t)k], 0);
        if (de == hipSuccess) de = hipMemcpyAsync((char *)out + px0 * os, t_out + px0 * os, npx * os, hipMemcpyDeviceToHost, c->s_out);
    };
    std::mutex mu;
    std::condition_variable cv;
    long long launched = 0;
    bool stop = false;
    bool threaded = nchunks > 2;
    std::thread downloader;
    if (threaded) try {
        downloader = std::thread([&] {
            de = hipSetDevice(c->device);
            for (long long k = 0; k < nchunks; ++k) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return launched > k || stop; });
                    if (launched <= k) return;
                }
                drain(k);
            }
        });
    } catch (...) {  // no exception crosses the ABI: fall back to issuing the downloads from this thread
        threaded = false;
    }
    for (long long k = 0; k < nchunks && e == hipSuccess; ++k) {
        const long long l0 = k * lpc, l1 = std::min((long long)lines, l0 + lpc);
        const size_t px0 = (size_t)l0 * samples, npx = (size_t)(l1 - l0) * samples;
        e = hipMemcpyAsync(t_in + px0 * es, (const char *)sigma0 + px0 * es, npx * es, hipMemcpyHostToDevice, c->s_in);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ready[(size_t)k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(ready[(size_t)k], c->s_in);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, ready[(size_t)k], 0);
        if (e == hipSuccess) e = launch(t_in + px0 * es, t_out + px0 * os, l1 - l0);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&done[(size_t)k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(done[(size_t)k], c->stream);
        if (threaded) {
            if (e == hipSuccess) {
                std::lock_guard<std::mutex> lk(mu);
                launched = k + 1;
            }
            cv.notify_one();
        } else if (e == hipSuccess && k > 0) drain(k - 1);
    }
    if (threaded) {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_one();
        downloader.join();
    } else if (e == hipSuccess && nchunks > 0) drain(nchunks - 1);
    hipError_t s1 = hipStreamSynchronize(c->s_in), s2 = hipStreamSynchronize(c->stream), s3 = hipStreamSynchronize(c->s_out);
    for (hipEvent_t ev : done) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : ready) if (ev) (void)hipEventDestroy(ev);
    if (c->arena_cap > XSW_ARENA_KEEP) {
        (void)hipFree(c->arena);
        c->arena = nullptr;
        c->arena_cap = 0;
    }
    for (hipError_t x : {de, s1, s2, s3}) if (e == hipSuccess) e = x;
    if (e != hipSuccess) return fail(c, XSW_EHIP, "detrend failed: %s", hipGetErrorString(e));
    return XSW_OK;
}

// ---------------------------------------------------------------------------------------- cross-pol noise flattening
template <typename T>
static hipError_t launch_nesz(hipStream_t s, const void *noise, const void *inc, void *scratch, double *out, long long lines,
                              long long samples, int nb, long long lpb)
{
    NeszPartial *part = (NeszPartial *)scratch;
    double *col = (double *)((char *)scratch + (size_t)nb * samples * sizeof(NeszPartial));
    double *x0 = col + 2 * samples;
    const unsigned gx = (unsigned)((samples + 255) / 256);
    hipLaunchKernelGGL((k_nesz_colsum<T>), dim3(gx, (unsigned)nb), dim3(256), 0, s, (const T *)noise, (const T *)inc, part, lines, samples, lpb);
    hipLaunchKernelGGL(k_nesz_colmean, dim3(gx), dim3(256), 0, s, part, col, samples, nb);
    hipLaunchKernelGGL(k_nesz_center, dim3(1), dim3(1024), 0, s, col, x0, samples);
    hipLaunchKernelGGL((k_nesz_rows<T>), dim3((unsigned)((lines + XSW_NESZ_LINES - 1) / XSW_NESZ_LINES)), dim3(256), 0, s, (const T *)noise, col, x0, out,
                       lines, samples);
    return hipGetLastError();
}

extern "C" int xsw_nesz_flatten(xsw_ctx *c, int64_t lines, int64_t samples, int32_t dtype, int32_t mem, const void *noise,
                                const void *inc, double *out)
{
    if (!c) return XSW_EINVAL;
    if (lines < 0 || samples < 0 || !noise || !inc || !out) return fail(c, XSW_EINVAL, "bad nesz_flatten argument");
    if (dtype != XSW_F32 && dtype != XSW_F64) return fail(c, XSW_EINVAL, "dtype must be XSW_F32 or XSW_F64");
    if (mem != XSW_MEM_HOST && mem != XSW_MEM_DEVICE) return fail(c, XSW_EINVAL, "bad mem kind");
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
    const size_t scratch_bytes = (size_t)nb * samples * sizeof(NeszPartial) + (2 * (size_t)samples + 8) * sizeof(double);
    void *scratch = nullptr;
    HIPCHK(c, hipMalloc(&scratch, scratch_bytes));
    hipError_t e = hipSuccess;
    const void *d_noise = noise, *d_inc = inc;
    double *d_out = out;
    void *stage = nullptr;
    if (mem == XSW_MEM_HOST) {
        const size_t in_b = ((size_t)n * es + 255) & ~(size_t)255;
        e = hipMalloc(&stage, 2 * in_b + (size_t)n * 8);
        if (e == hipSuccess) {
            d_noise = stage; d_inc = (char *)stage + in_b; d_out = (double *)((char *)stage + 2 * in_b);
            e = hipMemcpyAsync((void *)d_noise, noise, (size_t)n * es, hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync((void *)d_inc, inc, (size_t)n * es, hipMemcpyHostToDevice, c->stream);
        }
    }
    if (e == hipSuccess)
        e = dtype == XSW_F32 ? launch_nesz<float>(c->stream, d_noise, d_inc, scratch, d_out, lines, samples, (int)nb, lpb)
                             : launch_nesz<double>(c->stream, d_noise, d_inc, scratch, d_out, lines, samples, (int)nb, lpb);
    if (mem == XSW_MEM_HOST && e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream);
    // the scratch (and the staging area) are call-local: wait for the stream before freeing them
    hipError_t se = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = se;
    (void)hipFree(scratch);
    if (stage) (void)hipFree(stage);
    if (e != hipSuccess) return fail(c, e == hipErrorOutOfMemory ? XSW_ENOMEM : XSW_EHIP, "nesz_flatten failed: %s", hipGetErrorString(e));
    return XSW_OK;
}
