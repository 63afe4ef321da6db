/* libxsw -- MI355X (gfx950) wind-inversion hot path of xsarsea, flat C ABI.
 *
 * The reference (umr-lops/xsarsea) is pure Python; its "native" layer for this path is what numba
 * JIT-compiles at run time.  This header is the boundary a maintainer binds with ctypes in place of
 * those JIT kernels.  Each entry point names the reference interface it
 * replaces (paths relative to the reference's src/xsarsea/).
 *
 * Conventions
 *   - every function returns 0 on success, a negative XSW_E* code otherwise; the message is
 *     available from xsw_last_error(ctx);  no exception crosses the ABI;
 *   - pointers are caller-owned; `mem` says whether raster buffers are host or device pointers
 *     (device pointers must belong to the context's device);  LUT/axis pointers are always host;
 *   - rasters are flat, C-contiguous, `lines*samples` pixels; pixels are independent
 *     (the reference's gufunc core dimension "(n)" is only a loop, windspeed/windspeed.py:190);
 *   - complex values are interleaved (re, im);
 *   - a context is bound to one device and one stream; calls on one context are not thread-safe,
 *     different contexts are independent (one process per GPU uses one context; one process driving several GPUs
 *     uses one context per GPU from one host thread each -- xsarsea_amd.options.devices does exactly that).
 *
 * Host rasters (XSW_MEM_HOST) travel through a context-owned ring of page-locked staging buffers filled and drained by
 * host worker threads (XSW_HOST_THREADS, default 12, at most 32), one HIP stream per worker: chunk k is staged, uploaded,
 * inverted, downloaded and written to the caller's output while the other workers do the same for other chunks.  What
 * comes down the link is the answer as 4-byte GRID CODES (see xsw_invert_args.out_code_co), expanded to complex values
 * on the host from the same tables and by the same IEEE operations as on the device: bit-identical, 4 instead of 8 / 16
 * bytes per pixel and output over PCIe.
 */
#ifndef XSW_H
#define XSW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XSW_VERSION 4

enum { XSW_F32 = 0, XSW_F64 = 1 };           /* raster element type (complex rasters: c64 / c128) */
enum { XSW_MEM_HOST = 0, XSW_MEM_DEVICE = 1,
       XSW_MEM_HOST_PINNED = 2 /* host rasters in page-locked memory (xsw_host_alloc / hipHostMalloc): DMA reads them directly,
                                  the staging copy of XSW_MEM_HOST is skipped for the inputs */,
       XSW_MEM_DEVICE_SIGMA0_HOST = 3 /* XSW_VERSION >= 3, xsw_invert only: every raster lives in device memory EXCEPT sigma0_co /
                                  sigma0_cr, which are host rasters (pageable).  They travel through the workers' staging ring in
                                  row chunks -- the `stage` callback may fill a chunk itself: the bit-parity route converts
                                  sigma0 to dB there with numpy's own float32 log10 (windspeed.py:126-130) -- and each chunk is
                                  searched on its worker's stream against the resident incidence / ancillary rasters, results
                                  written in place.  Synchronous.  The context's stream is synchronised first. */ };
enum {
    XSW_ALGO_AUTO = 0,       /* pruned when the LUT axes are uniform and finite, else exact        */
    XSW_ALGO_PRUNED = 1,     /* exact branch-and-bound search (production kernel)                   */
    XSW_ALGO_EXHAUSTIVE = 2, /* full (wspd x phi) sweep, LUT slice tiled through LDS; float32 screening,
                                float64 settle (results identical to the reference's)                */
    XSW_ALGO_EXACT = 3,      /* full sweep in the reference's operation order (slow, any LUT)       */
    XSW_ALGO_EXHAUSTIVE_F64 = 4 /* the LDS-tiled sweep with float64 screening                       */
};
enum {
    XSW_OK = 0,
    XSW_EINVAL = -1,   /* bad argument */
    XSW_EHIP = -2,     /* HIP runtime error */
    XSW_ENOLUT = -3,   /* required LUT not uploaded */
    XSW_ENOMEM = -4
};

typedef struct xsw_ctx xsw_ctx;

/* A model LUT in dB, as produced by the reference's Model.to_lut(units="dB") (windspeed/models.py:186-230)
 * but kept incidence-major: co-pol db[n_inc][n_wspd][n_phi], cross-pol db[n_inc][n_wspd] (n_phi = 0).
 * Replaces the closure arrays of _invert_from_model_numpy (windspeed/windspeed.py:144-181).
 *
 * The remaining tables are OPTIONAL (NULL = filled by the library with the host libm).  They exist so
 * that a caller can hand over the values ITS math library produces for the handful of transcendental
 * expressions of the reference whose last bit is platform dependent (numpy dispatches SIMD variants
 * of cos/sin/arctan2/abs), making the device results bit-identical to that caller's CPU path:
 *   cos_phi, sin_phi [n_phi]        cos/sin(radians(phi)) of the candidate vectors (windspeed.py:167-168)
 *   out_dir [2][n_phi][2]           exp(1j*deg2rad(+phi)) and exp(1j*deg2rad(-phi)) as (re, im) (:235-236, :247)
 *   abs_co  [n_wspd][n_phi]         abs(wspd*exp(1j*deg2rad(phi)))  (np.abs(wind_co), :257-274)
 *   dual_dir [2][n_wspd][n_phi][2]  exp(1j*angle(sol)), exp(1j*angle(sol_2)) as (re, im) (:270-276)     */
typedef struct {
    const double *db;
    const double *inc;
    const double *wspd;
    const double *phi;
    const double *cos_phi;
    const double *sin_phi;
    const double *out_dir;
    const double *abs_co;
    const double *dual_dir;
    int32_t n_inc, n_wspd, n_phi;
} xsw_lut;

/* Arguments of one inversion call == the five gufunc inputs and two outputs of
 * __invert_from_model_1d (windspeed/windspeed.py:183-282, wrapper :306-323), plus the sigma0->dB
 * conversion of invert_from_model (:126-130) and the dual-pol select (:426-428) fused on request. */
typedef struct {
    int64_t lines, samples;
    int32_t dtype;          /* XSW_F32 / XSW_F64: inc, sigma0_co, sigma0_cr, dsig_cr; anc is complex of it */
    int32_t out_dtype;      /* XSW_F32 -> complex64 outputs, XSW_F64 -> complex128 (reference)      */
    int32_t mem;            /* XSW_MEM_HOST / XSW_MEM_DEVICE, applies to every raster pointer below   */
    int32_t sigma0_is_db;   /* 0: linear sigma0, converted as 10*log10(x + 1e-15) in `dtype` arithmetic
                               (:126-130);  1: already dB                                             */
    int32_t algo;           /* XSW_ALGO_*                                                             */
    int32_t dual_select;    /* 1: out_cr receives where(|co|<5 or |dual|<5, co, dual) (:426-428)      */
    const void *inc;        /* incidence, degrees                                                     */
    const void *sigma0_co;  /* NULL: no co-pol search (cross-pol only)                                */
    const void *sigma0_cr;  /* NULL: no cross-pol search                                              */
    const void *dsig_cr;    /* NULL: use dsig_cr_scalar (:122-123)                                    */
    const void *anc;        /* ancillary wind, complex, antenna convention; NULL: all NaN             */
    double dsig_co;         /* :24 default 0.1                                                        */
    double dsig_cr_scalar;
    void *out_co;           /* complex; NULL allowed when sigma0_co is NULL                           */
    void *out_cr;           /* complex; NULL: not written                                             */
    int32_t *out_idx;       /* optional int32[n][3] = (i_wspd, i_phi, i_wspd_cr), -1 where no search  */
    /* XSW_VERSION >= 2.  Optional: the answer as grid codes, 4 bytes per pixel and search -- the retrieved wind IS a grid
     * point of the LUT (windspeed.py:228-232, :269), so this is the whole result; xsw_expand_codes turns codes into the
     * complex values out_co / out_cr would have received, bit for bit.  When a code pointer is given the corresponding
     * complex output may be NULL (nothing else is written for that search): 4 B/px instead of 8 or 16 to store, gather
     * over xGMI (multi-GPU) or move over PCIe.
     *   co code: i_wspd * n_phi + i_phi in bits 0..29, bit 30 set when the -phi solution was chosen (:234-242);
     *            XSW_CODE_NAN_RE = (nan, 0) [incidence NaN, :198-201 / ancillary NaN, :204-207], XSW_CODE_NAN = (nan, nan) [:250]
     *   cr code: i_wspd_cr in bits 0..29 (XSW_CODE_NO_INDEX: no cross-pol search ran, wind_dual = (nan, nan) :278), bit 30
     *            (XSW_CODE_PICK_CO) set when dual_select picked the co-pol wind; XSW_CODE_NAN_RE as above.                  */
    uint32_t *out_code_co;
    uint32_t *out_code_cr;
    /* Optional caller-side staging (host rasters only; ignored for XSW_MEM_DEVICE).  When non-NULL, a worker thread that is
     * about to stage pixels [px0, px0 + npx) of input raster `which` (0 inc, 1 sigma0_co, 2 sigma0_cr, 3 dsig_cr, 4 anc) calls
     *     stage(stage_user, which, px0, npx, dst)
     * with `dst` = the page-locked staging area for that piece (npx elements of `dtype`, complex for anc).  Return 1 when the
     * callback has filled dst (the library then does not read the raster pointer, which must still be non-NULL to request the
     * search), 0 to let the library copy from the raster pointer as usual, < 0 to abort the call (XSW_EINVAL is returned).
     * Called concurrently from several threads, for disjoint pieces.  This is how the Python layer runs numpy's own float32
     * log10 (the reference's sigma0 -> dB arithmetic, :126-130) inside the pipeline instead of in a pass of its own. */
    int (*stage)(void *stage_user, int32_t which, int64_t px0, int64_t npx, void *dst);
    void *stage_user;
} xsw_invert_args;

#define XSW_CODE_NAN_RE  0xFFFFFFFFu
#define XSW_CODE_NAN     0xFFFFFFFEu
#define XSW_CODE_PICK_CO  0x40000000u
#define XSW_CODE_NO_INDEX 0x3FFFFFFFu

/* Evaluated-work counters of the last xsw_invert call with stats enabled. */
typedef struct {
    uint64_t pixels_co;        /* pixels that ran a co-pol search                   */
    uint64_t cand_co;          /* co-pol candidates actually scored                 */
    uint64_t pixels_exact;     /* pixels that took the exact full-scan path         */
    uint64_t pixels_cr;        /* pixels that ran a cross-pol search (XSW_ALGO_EXHAUSTIVE: pixels the float32
                                  sweep could not decide, finished by the float64 box search) */
} xsw_stats;

int xsw_version(void);
int xsw_device_count(void);

int xsw_ctx_create(int device, xsw_ctx **ctx);
int xsw_ctx_destroy(xsw_ctx *ctx);
const char *xsw_last_error(const xsw_ctx *ctx); /* ctx may be NULL: last creation error */

/* Launch on a caller-provided hipStream_t (e.g. torch's current stream; NULL = the default stream).
 * A new context launches on a private non-blocking stream; xsw_use_own_stream() returns to it.
 * XSW_MEM_DEVICE buffers are read and written in stream order ON THAT STREAM ONLY: a caller whose producers run on another
 * stream (torch's, say) must either hand that stream over here or synchronise it before xsw_invert / xsw_detrend / ... */
/* Changing the stream orders the new stream after the work the context queued on the previous one (an event the new stream
 * waits for; the host does not block): the work list that hands pixels from k_invert_band to k_invert_list and the scratch of
 * xsw_nesz_flatten are context-owned and reused by the next call. */
int xsw_set_stream(xsw_ctx *ctx, void *hip_stream);
int xsw_use_own_stream(xsw_ctx *ctx);
int xsw_synchronize(xsw_ctx *ctx);

/* Replaces Model.to_lut(...) -> closure arrays (windspeed.py:144-181).  Either may be NULL (kept).
 * Device memory held per context for the default co-pol table (501 x 499 x 181): 368 MB float64 + 184 MB float32 copies,
 * 363 MB transposed copy, 394 MB inverse-row table (first row of every direction at or above each of 2048 dB thresholds
 * per incidence slice: what the band search reads instead of bisecting), 6 MB of block / band {min, max} tables (block
 * pyramid), 6 MB of tail minima, ~10 MB of small tables; all derived copies are produced on the device at install (a few ms). */
int xsw_lut_upload(xsw_ctx *ctx, const xsw_lut *co, const xsw_lut *cr);

/* Replaces _invert_from_model_numpy (windspeed.py:132-331).  Asynchronous on the context's stream
 * when mem == XSW_MEM_DEVICE; synchronous (returns with outputs filled) for host memory.
 * XSW_ALGO_PRUNED on a LUT whose columns rise monotonically with wind speed (every built-in GMF over most of its rows) runs
 * as FOUR launches on one stream:
 *   k_invert_band    decides the pixels its band rule can (windows inside the monotone rows, short runs of band rows; wide windows
 *                    narrowed to the directions in which the band can meet the disc, from the inverse-row table); hands the
 *                    pixels whose band holds a long run of rows along the a-priori direction (XSW_LONG_RUN = 5 or more), or a
 *                    window that reaches past the monotone rows by a tail it can sweep, to list B as 48-byte records;
 *   k_invert_band2   list B: per record a bound from the sigma0 contour itself (inverse-row table), the live arc of directions,
 *                    per direction the joint shrink of band and chord, batched sweep;
 *   k_invert_blocks  list C: the finite pixels the band rule is not for (windows past the monotone rows, bands of thousands of
 *                    candidates, sigma0 outliers): branch-and-bound over min / max tables of LUT cells (32 x 32 candidates), blocks
 *                    (4 x 16) and quarter blocks (4 x 4), both cost terms bound together, four pixels per wave;
 *   k_invert_list    list G: the rest (non-finite inputs, near-ties, whatever overflowed), the general algorithm.
 * The work lists are owned by the context and sized by the largest raster seen (n pixels): list G n/8 entries, lists B and C
 * n/2 entries each (4 bytes per entry), list B's records 48 bytes x n/2, two strip masks of one bit per pixel -- 28.8 bytes per
 * pixel in all (11.5 GB for a 20000 x 20000 raster; HBM holds 288 GB).  A list that overflows is continued in its strip mask:
 * the consumer takes the list, then exactly the marked pixels (stage 1 is redone for those of list B).  If the lists cannot be
 * allocated the one-kernel path runs; any other LUT, and XSW_ALGO_EXACT, take the general kernel (k_invert).
 * Results do not depend on the route.  Environment switches for A/B measurements and the tests of every route: XSW_LONG_RUN=0
 * (no k_invert_band2), XSW_NO_BLOCKS_KERNEL=1, XSW_NO_BAND=1 (general kernel only), XSW_NO_RECORDS=1, XSW_NO_STRIP_MASKS=1,
 * XSW_B2_REFINE_MIN / XSW_B2_AREA / XSW_B2_ROWS_MAX / XSW_TAIL_SWEEP (routing thresholds: INTEGRATION.md). */
int xsw_invert(xsw_ctx *ctx, const xsw_invert_args *args);

/* Grid codes -> complex winds (the store of __invert_from_model_1d: wind_co = wspd * exp(1j * deg2rad(+-phi)) :235-247,
 * wind_dual = wspd_dual * exp(1j * angle(wind_co)) :270-276), from the tables of the context's current LUTs.  code_co may be
 * NULL when only cross-pol codes exist (cross-pol-only inversion: wind_dual = wspd_dual + 0j), code_cr / out_cr may be NULL.
 * out_dtype XSW_F32 -> complex64, XSW_F64 -> complex128.  mem = XSW_MEM_DEVICE: one HBM-bound kernel, asynchronous on the
 * context's stream; XSW_MEM_HOST: on the calling thread.  Both give the bits xsw_invert writes to out_co / out_cr. */
int xsw_expand_codes(xsw_ctx *ctx, int64_t n, int32_t mem, int32_t out_dtype, const uint32_t *code_co,
                     const uint32_t *code_cr, void *out_co, void *out_cr);

/* XSW_VERSION >= 3.  The device form of xsw_expand_codes on a stream of the caller's choice (a HIP stream handle), without
 * touching the context's launch stream: a consumer that receives codes piece by piece (the row chunks of a multi-GPU gather)
 * expands each piece on a side stream as it lands, next to the inversions still queued on the launch stream. */
int xsw_expand_codes_on_stream(xsw_ctx *ctx, void *stream, int64_t n, int32_t out_dtype, const uint32_t *code_co,
                               const uint32_t *code_cr, void *out_co, void *out_cr);

/* Page-locked host memory for rasters a caller fills itself (XSW_MEM_HOST_PINNED); freed by xsw_host_free or with the context. */
int xsw_host_alloc(xsw_ctx *ctx, size_t bytes, void **out);
int xsw_host_free(xsw_ctx *ctx, void *p);
/* Host worker threads of the XSW_MEM_HOST paths (0 = default: XSW_HOST_THREADS or 12; at most 32).  Every worker keeps one
 * page-locked staging buffer and one device buffer of a chunk between calls (~2 Mpx: 40 MB each for float32 mono rasters,
 * ~110 MB for float64 dual-pol), so a context holds up to threads x chunk of pinned host memory; after every host-memory call
 * what exceeds XSW_STAGING_KEEP_MB (environment, default 1536 MB per context, a worker counted with the larger of its page-locked
 * and its device buffer: the device side also holds the chunk's work lists and records) is released again, and lowering the thread count
 * frees the workers that are no longer used at once. */
int xsw_set_host_threads(xsw_ctx *ctx, int n);

/* Enable (1) / disable (0) device-side work counters; read them after synchronising.  on = 1: XSW_ALGO_PRUNED runs its statistics
 * instantiation (k_invert_band sweeps every window itself and counts every candidate it scores: cand_co is what the band rule
 * leaves of the grid).  XSW_VERSION >= 4, on = 2: the PRODUCTION chain runs unchanged and the kernels behind k_invert_band count
 * what THEY score (xsw_stats_read_chain): measured candidates per second of k_invert_band2 / k_invert_blocks / k_invert_list. */
int xsw_stats_enable(xsw_ctx *ctx, int on);
int xsw_stats_read(xsw_ctx *ctx, xsw_stats *out);
typedef struct {
    uint64_t cand_band2;      /* candidates scored by k_invert_band2 (after contour bound and joint shrink)          */
    uint64_t cand_blocks;     /* candidates scored by k_invert_blocks (16 per swept quarter block)                    */
    uint64_t cand_list;       /* candidates scored by k_invert_list                                                   */
    uint64_t pixels_refined;  /* list-B records k_invert_band2 put through its refinement (contour bound, live arc)  */
} xsw_chain_stats;
int xsw_stats_read_chain(xsw_ctx *ctx, xsw_chain_stats *out);

/* Measurement aid (no reference counterpart): while enabled, every XSW_ALGO_PRUNED inversion of DEVICE rasters that takes the
 * four-kernel chain (k_invert_band, k_invert_band2, k_invert_blocks, k_invert_list: see xsw_invert) is bracketed by HIP events on the
 * launch stream, one between every two kernels.  xsw_timing_read synchronises, returns the summed durations since the last read and
 * forgets them. */
typedef struct {
    int64_t launches;        /* inversions measured                                                   */
    double first_kernel_ms;  /* k_invert_band, summed over the launches                               */
    double second_kernel_ms; /* k_invert_list, summed over the launches                               */
    int64_t last_list_pixels;/* pixels the most recent launch left to k_invert_list                   */
    double band2_kernel_ms;  /* k_invert_band2 (the pixels with long runs of band rows, between the two), summed   */
    int64_t last_band2_pixels;/* pixels the most recent launch handed to k_invert_band2              */
    double blocks_kernel_ms; /* XSW_VERSION >= 3: k_invert_blocks (block pyramid, after k_invert_band2), summed */
    int64_t last_blocks_pixels;/* pixels the most recent launch handed to k_invert_blocks             */
} xsw_timing;
int xsw_timing_enable(xsw_ctx *ctx, int on);
int xsw_timing_read(xsw_ctx *ctx, xsw_timing *out);

/* Replaces the low->high resolution interpolation of Model._normalize_lut (windspeed/models.py:142-168:
 * `lut.interp(incidence=, wspd=, phi=)`, i.e. three sequential linear 1-D interpolations in the order
 * incidence -> wspd -> phi) on the device, with the arithmetic of scipy.interp1d
 * (slope = (y_hi-y_lo)/(x_hi-x_lo); y = slope*(x_new-x_lo) + y_lo) so that the result is bit-identical.
 * raw[n_inc_raw][n_wspd_raw][n_phi_raw] and out[n_inc][n_wspd][n_phi] are host pointers (n_phi* = 0 for a
 * cross-pol table).  ValueError-equivalent (XSW_EINVAL) when a target point lies outside the raw axis
 * (bounds_error=True) or an axis is not strictly ascending. */
int xsw_lut_interp(xsw_ctx *ctx, const double *raw, const double *inc_raw, const double *wspd_raw,
                   const double *phi_raw, int32_t n_inc_raw, int32_t n_wspd_raw, int32_t n_phi_raw,
                   const double *inc, const double *wspd, const double *phi, int32_t n_inc, int32_t n_wspd,
                   int32_t n_phi, double *out);

/* Device-side LUT preparation for the built-in GMFs (gmf_id: XSW_GMF_* below): evaluates the model on its RAW grid
 * (GmfModel._raw_lut, windspeed/gmfs.py:350-395), brings it to the TARGET grid with the three sequential linear
 * interpolations of Model._normalize_lut (windspeed/models.py:142-168; skipped when the two grids are equal, e.g.
 * resolution="low"), converts to dB (models.py:210-216) and installs the result as the context's co-pol (phi axes given)
 * or cross-pol (n_phi_raw == 0, target->n_phi == 0) LUT, exactly as xsw_lut_upload would -- without the table visiting
 * the host.  `target` carries the target axes and the optional host tables of xsw_lut; target->db is ignored.
 * Values agree with the host-built LUT to ~1e-13 dB (device libm), not bit for bit: use xsw_lut_upload with a
 * host-built table when bit parity with a CPU run is wanted. */
int xsw_lut_build(xsw_ctx *ctx, int32_t gmf_id, const double *inc_raw, int32_t n_inc_raw, const double *wspd_raw,
                  int32_t n_wspd_raw, const double *phi_raw, int32_t n_phi_raw, const xsw_lut *target);

/* Copies the context's current co-pol (cross == 0: out_db[n_inc][n_wspd][n_phi]) or cross-pol (cross != 0:
 * out_db[n_inc][n_wspd]) dB table back to the host, unpadded -- what Model.to_lut(units="dB") would hold
 * (windspeed/models.py:186-230) for a LUT that was built on the device.  Synchronous. */
int xsw_lut_read(xsw_ctx *ctx, int32_t cross, double *out_db);

/* Built-in analytic GMFs on the device: out[i] = gmf(inc[i], wspd[i], phi[i]) over n already-broadcast float64
 * elements (phi may be NULL for cross-pol models).  Replaces the numba-vectorised forward GMF of
 * GmfModel.__call__(..., broadcast=True) (windspeed/gmfs.py:202-214, :293-316) for the models of gmfs_impl.py.
 * gmf_id: XSW_GMF_* below.  Values agree with the host evaluation to ~1e-14 relative (device libm), not bitwise. */
enum {
    XSW_GMF_CMOD5 = 0, XSW_GMF_CMOD5N = 1, XSW_GMF_CMOD5N_PR_ZHANGA = 2, XSW_GMF_CMOD5N_PR_MOUCHE1 = 3,
    XSW_GMF_CMODIFR2 = 4, XSW_GMF_RS2_V2 = 5, XSW_GMF_S1_V2 = 6, XSW_GMF_RCM_NOAA = 7, XSW_GMF_S1_V3_EW_REC = 8,
    XSW_GMF_RS2_V3 = 9, XSW_GMF_RCM_V3 = 10, XSW_GMF_RCM_V4 = 11, XSW_GMF_RS2_V4 = 12
};
int xsw_gmf_eval(xsw_ctx *ctx, int32_t gmf_id, int64_t n, int32_t mem, const double *inc, const double *wspd,
                 const double *phi, double *out);

/* Replaces the per-pixel part of sigma0_detrend (detrend.py:63-64):
 * out[l][s] = sigma0[l][s] / ratio_row[s], ratio_row = g / nanmean(g) (float64, host pointer).
 * out is float64 (the reference's result dtype) when out_dtype == XSW_F64.  Host rasters: synchronous; device
 * rasters: asynchronous on the context's stream (ratio_row is consumed before the call returns). */
int xsw_detrend(xsw_ctx *ctx, int64_t lines, int64_t samples, int32_t dtype, int32_t out_dtype, int32_t mem,
                const void *sigma0, const double *ratio_row, void *out);

/* Replaces nesz_flattening (windspeed/utils.py:94-163), the full-raster pass in front of the dual-pol inversion:
 * out[l][s] = 10 ** ((inc_row[s] * slope_l + icpt_l - 1) / 10), with inc_row = nanmean(inc, axis 0), and (slope_l, icpt_l)
 * the degree-1 least-squares fit of 10*log10(noise[l], NaNs replaced by the column nan-mean) against inc_row over the
 * finite samples of line l; a line without any finite sample is NaN.  noise/inc are `dtype` rasters (host or device per
 * `mem`), out is float64 (the reference's result dtype).  Sums are float64 and the fit is closed-form: agrees with
 * numpy's polyfit-based result to ~1e-13 relative for float64 rasters (float32 rasters: float32 dB arithmetic like the
 * reference's own, 1e-5).  Device rasters: asynchronous on the context's stream (scratch is context-owned); host rasters:
 * returns with `out` filled. */
int xsw_nesz_flatten(xsw_ctx *ctx, int64_t lines, int64_t samples, int32_t dtype, int32_t mem, const void *noise,
                     const void *inc, double *out);

#ifdef __cplusplus
}
#endif
#endif /* XSW_H */
