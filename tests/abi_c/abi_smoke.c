/* Plain-C caller of libxsw (no Python, no torch): the drop-in boundary is a flat C ABI.
 * Reads a problem from a binary file written by tests/test_gpu_api.py, runs xsw_invert on host buffers and writes the
 * complex128 winds + indices back.  File layout (little endian):
 *   int32 n_inc, n_wspd, n_phi, n_wcr, n_pix;  double dsig_co;
 *   double inc_ax[n_inc], w_ax[n_wspd], phi_ax[n_phi], co[n_inc*n_wspd*n_phi], wcr_ax[n_wcr], cr[n_inc*n_wcr];
 *   double inc[n_pix], s_co_db[n_pix], s_cr_db[n_pix], dsig_cr[n_pix], anc[2*n_pix]
 * Output: double out_co[2*n_pix], out_cr[2*n_pix]; int32 idx[3*n_pix]. */
#include <stdio.h>
#include <stdlib.h>
#include "xsw.h"

static double *rd(FILE *f, size_t n)
{
    double *p = (double *)malloc((n ? n : 1) * sizeof(double));
    if (!p || fread(p, sizeof(double), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return p;
}

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: abi_smoke problem.bin result.bin\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    int32_t h[5];
    double dsig_co;
    if (fread(h, sizeof(int32_t), 5, f) != 5 || fread(&dsig_co, sizeof(double), 1, f) != 1) return 2;
    const size_t ni = h[0], nw = h[1], np = h[2], nc = h[3], n = h[4];
    double *inc_ax = rd(f, ni), *w_ax = rd(f, nw), *phi_ax = rd(f, np), *co = rd(f, ni * nw * np);
    double *wcr_ax = rd(f, nc), *cr = rd(f, ni * nc);
    double *inc = rd(f, n), *sco = rd(f, n), *scr = rd(f, n), *dsig = rd(f, n), *anc = rd(f, 2 * n);
    fclose(f);

    xsw_ctx *ctx = NULL;
    if (xsw_ctx_create(0, &ctx) != XSW_OK) { fprintf(stderr, "ctx: %s\n", xsw_last_error(NULL)); return 1; }
    xsw_lut lco = {co, inc_ax, w_ax, phi_ax, NULL, NULL, NULL, NULL, NULL, (int32_t)ni, (int32_t)nw, (int32_t)np};
    xsw_lut lcr = {cr, inc_ax, wcr_ax, NULL, NULL, NULL, NULL, NULL, NULL, (int32_t)ni, (int32_t)nc, 0};
    if (xsw_lut_upload(ctx, &lco, &lcr) != XSW_OK) { fprintf(stderr, "lut: %s\n", xsw_last_error(ctx)); return 1; }

    double *out_co = (double *)malloc(2 * n * sizeof(double)), *out_cr = (double *)malloc(2 * n * sizeof(double));
    int32_t *idx = (int32_t *)malloc(3 * n * sizeof(int32_t));
    xsw_invert_args a = {0};
    a.lines = 1; a.samples = (int64_t)n;
    a.dtype = XSW_F64; a.out_dtype = XSW_F64; a.mem = XSW_MEM_HOST; a.sigma0_is_db = 1; a.algo = XSW_ALGO_AUTO;
    a.inc = inc; a.sigma0_co = sco; a.sigma0_cr = scr; a.dsig_cr = dsig; a.anc = anc;
    a.dsig_co = dsig_co; a.dsig_cr_scalar = 0.1;
    a.out_co = out_co; a.out_cr = out_cr; a.out_idx = idx;
    if (xsw_invert(ctx, &a) != XSW_OK) { fprintf(stderr, "invert: %s\n", xsw_last_error(ctx)); return 1; }
    /* error behaviour: a NULL incidence raster is refused with a message, nothing crashes */
    a.inc = NULL;
    if (xsw_invert(ctx, &a) != XSW_EINVAL || !xsw_last_error(ctx)[0]) { fprintf(stderr, "expected XSW_EINVAL\n"); return 1; }
    xsw_ctx_destroy(ctx);

    f = fopen(argv[2], "wb");
    if (!f) { perror("open out"); return 2; }
    fwrite(out_co, sizeof(double), 2 * n, f);
    fwrite(out_cr, sizeof(double), 2 * n, f);
    fwrite(idx, sizeof(int32_t), 3 * n, f);
    fclose(f);
    printf("abi_smoke ok: %zu pixels, libxsw version %d\n", n, xsw_version());
    return 0;
}
