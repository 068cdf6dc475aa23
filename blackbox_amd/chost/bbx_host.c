/* bbx_host.c -- host-side helpers of the overscan solve (plain C, no GPU): the per-column
 * sigma-clipped statistics of the horizontal-overscan strip and the flat clipped
 * statistics, written out with exactly the float operations (and their order) of the
 * numpy code in blackbox_amd/overscan.py, which in turn follows the reference
 * (blackbox.py:6649-6659, astropy sigma_clip).  Build: gcc -O2 -ffp-contract=off.
 * tests/test_host_overscan.py holds both against the numpy versions bit for bit. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* overscan.hos_column_stats(data_hos, mask_hos, accum='f32seq'):
 * data [nrow][ncol] float32, mask [nrow][ncol] uint8 (non-zero = masked)
 * -> n [ncol] int64, mean [ncol] float32, std [ncol] float32 */
/* (loads are unconditional and the selects written on values, so that gcc if-converts and
 * vectorises the column loops; the clones pick AVX2 at load time where the CPU has it) */
__attribute__((target_clones("avx2", "default")))
int bbx_hos_column_stats_f32seq(const float *restrict data, const uint8_t *restrict mask, int nrow, int ncol,
                                int64_t *restrict n_out, float *restrict mean_out, float *restrict std_out) {
    if (nrow < 1 || nrow > 4096 || ncol < 1) return -1;
    /* Rows outside, columns inside: every column still sees its rows in order 0..nrow-1 (the
     * order of numpy's reduction over axis 0), and the inner loops run along memory.
     * Masks are kept as 0.0 / 1.0 doubles (cur, ok) next to the data converted once. */
    const size_t nc = (size_t)ncol, np_ = (size_t)nrow * nc;
    /* scratch kept per thread between calls (a few hundred KB: malloc would mmap and fault it in every time) */
    static __thread void *scratch = 0;
    static __thread size_t scratch_bytes = 0;
    const size_t wbytes = (3 * np_ + 6 * nc) * sizeof(double), fbytes = (3 * nc + np_) * sizeof(float);
    if (scratch_bytes < wbytes + fbytes) {
        free(scratch);
        scratch = malloc(wbytes + fbytes);
        scratch_bytes = scratch ? wbytes + fbytes : 0;
        if (!scratch) return -3;
    }
    double *restrict w = (double *)scratch;
    float *restrict fw = (float *)((char *)scratch + wbytes);
    double *restrict dd = w, *restrict cur = w + np_, *restrict ok = w + 2 * np_;
    double *restrict lo = w + 3 * np_, *restrict hi = lo + nc, *restrict s = lo + 2 * nc, *restrict q = lo + 3 * nc,
           *restrict mean = lo + 4 * nc, *restrict cnt = lo + 5 * nc;
    for (size_t x = 0; x < nc; x++) { lo[x] = NAN; hi[x] = NAN; }
    for (size_t k = 0; k < np_; k++) {
        const double d = (double)data[k];
        const double fin = (d - d == 0.0) ? 1.0 : 0.0;             /* finite <=> d - d == 0 */
        const double v = mask[k] ? 0.0 : fin;
        dd[k] = d; ok[k] = v; cur[k] = v;
    }
    for (int it = 0; it < 5; it++) {
        for (size_t x = 0; x < nc; x++) { const double c = cur[x], d = dd[x]; s[x] = (c != 0.0) ? d : 0.0; cnt[x] = c; }
        for (int i = 1; i < nrow; i++) {
            const double *restrict dr = dd + (size_t)i * nc; const double *restrict cr = cur + (size_t)i * nc;
            for (size_t x = 0; x < nc; x++) {
                const double c = cr[x], d = dr[x];
                const double t = (c != 0.0) ? d : 0.0;
                s[x] = s[x] + t; cnt[x] = cnt[x] + c;
            }
        }
        for (size_t x = 0; x < nc; x++) mean[x] = s[x] / cnt[x];
        for (size_t x = 0; x < nc; x++) {
            const double c = cur[x], d = dd[x];
            const double dev0 = mean[x] - d, dev = (c != 0.0) ? dev0 : 0.0;
            q[x] = dev * dev;
        }
        for (int i = 1; i < nrow; i++) {
            const double *restrict dr = dd + (size_t)i * nc; const double *restrict cr = cur + (size_t)i * nc;
            for (size_t x = 0; x < nc; x++) {
                const double c = cr[x], d = dr[x];
                const double dev0 = mean[x] - d, dev = (c != 0.0) ? dev0 : 0.0;
                q[x] = q[x] + dev * dev;
            }
        }
        for (size_t x = 0; x < nc; x++) {
            const double sd = sqrt(q[x] / cnt[x]);
            const double nlo = mean[x] - 2.5 * sd, nhi = mean[x] + 2.5 * sd;
            const int upd = cnt[x] > 0.0;
            lo[x] = upd ? nlo : lo[x]; hi[x] = upd ? nhi : hi[x];
        }
        for (int i = 0; i < nrow; i++) {
            const double *restrict dr = dd + (size_t)i * nc; double *restrict cr = cur + (size_t)i * nc;
            for (size_t x = 0; x < nc; x++) {
                const double d = dr[x];
                const int in = (d >= lo[x]) & (d <= hi[x]);
                cr[x] = in ? cr[x] : 0.0;
            }
        }
    }
    /* final statistics in float32, rows in order */
    float *restrict tot = fw, *restrict tot2 = fw + nc, *restrict fm = fw + 2 * nc, *restrict okf = fw + 3 * nc;
    for (size_t x = 0; x < nc; x++) { tot[x] = 0.0f; tot2[x] = 0.0f; cnt[x] = 0.0; }
    for (int i = 0; i < nrow; i++) {
        const double *restrict dr = dd + (size_t)i * nc; const double *restrict orow = ok + (size_t)i * nc;
        const float *restrict fr = data + (size_t)i * nc; float *restrict of = okf + (size_t)i * nc;
        for (size_t x = 0; x < nc; x++) {
            const double d = dr[x];
            const int out = (d < lo[x]) | (d > hi[x]);
            const double o = out ? 0.0 : orow[x];
            const float f = fr[x];
            cnt[x] = cnt[x] + o;
            tot[x] = tot[x] + ((o != 0.0) ? f : 0.0f);
            of[x] = (float)o;
        }
    }
    for (size_t x = 0; x < nc; x++) fm[x] = tot[x] / (float)cnt[x];
    for (int i = 0; i < nrow; i++) {
        const float *restrict fr = data + (size_t)i * nc; const float *restrict of = okf + (size_t)i * nc;
        for (size_t x = 0; x < nc; x++) {
            const float dev0 = fr[x] - fm[x], dev = (of[x] != 0.0f) ? dev0 : 0.0f;
            tot2[x] = tot2[x] + dev * dev;
        }
    }
    for (size_t x = 0; x < nc; x++) {
        n_out[x] = (int64_t)cnt[x];
        mean_out[x] = fm[x];
        std_out[x] = sqrtf(tot2[x] / (float)(cnt[x] - 1.0));
    }
    return 0;
}

/* overscan.clipped_stats_flat(values, sigma, maxiters, accum='f32seq') for float32 input:
 * v [n] (finite values only are used) -> out[0] = mean, out[1] = std (both float32 values
 * stored as double), returns the number of survivors (or -1) */
int64_t bbx_clipped_stats_flat_f32seq(const float *values, int64_t n, double sigma, int maxiters, double *out) {
    float *v = (float *)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
    if (!v) return -1;
    int64_t m = 0;
    for (int64_t i = 0; i < n; i++) if (isfinite(values[i])) v[m++] = values[i];
    float mean = NAN, sd = NAN;
    for (int it = 0; it < maxiters; it++) {
        if (m == 0) break;
        float s = 0.0f;                                       /* np.cumsum(v, dtype=float32)[-1] */
        for (int64_t i = 0; i < m; i++) s = (i == 0) ? v[0] : s + v[i];
        mean = s / (float)m;
        float q = 0.0f;
        for (int64_t i = 0; i < m; i++) { const float d = v[i] - mean; const float t = d * d; q = (i == 0) ? t : q + t; }
        sd = sqrtf(q / (float)m);
        const float lo = (float)((double)mean - (double)sd * sigma), hi = (float)((double)mean + (double)sd * sigma);
        int64_t k = 0;
        for (int64_t i = 0; i < m; i++) if (v[i] >= lo && v[i] <= hi) v[k++] = v[i];
        if (k == m) break;
        m = k;
    }
    if (m > 0) {
        /* statistics of the survivors (recomputed like the numpy code does after the loop) */
        float s = 0.0f;
        for (int64_t i = 0; i < m; i++) s = (i == 0) ? v[0] : s + v[i];
        mean = s / (float)m;
        float q = 0.0f;
        for (int64_t i = 0; i < m; i++) { const float d = v[i] - mean; const float t = d * d; q = (i == 0) ? t : q + t; }
        sd = sqrtf(q / (float)m);
    }
    out[0] = (double)mean; out[1] = (double)sd;
    free(v);
    return m;
}

/* overscan.polyfit_exact, the part before LAPACK: rows of the cached Vandermonde matrix V
 * [n][order] picked by mask -> lhs [m][order] with every column divided by its Euclidean norm
 * (scale[order]; squares summed row by row like numpy's sum over axis 0).  Returns m. */
int64_t bbx_polyfit_prep(const double *restrict V, const uint8_t *restrict mask, int64_t n, int order,
                         double *restrict lhs, double *restrict scale) {
    if (order < 1 || order > 16 || n < 0) return -1;
    double acc[16];
    int64_t m = 0;
    for (int64_t i = 0; i < n; i++) {
        if (!mask[i]) continue;
        const double *restrict r = V + i * order;
        double *restrict o = lhs + m * order;
        if (m == 0) for (int j = 0; j < order; j++) { o[j] = r[j]; acc[j] = r[j] * r[j]; }
        else for (int j = 0; j < order; j++) { o[j] = r[j]; acc[j] = acc[j] + r[j] * r[j]; }
        m++;
    }
    if (m == 0) { for (int j = 0; j < order; j++) scale[j] = 0.0; return 0; }
    for (int j = 0; j < order; j++) scale[j] = sqrt(acc[j]);
    for (int64_t i = 0; i < m; i++) {
        double *restrict o = lhs + i * order;
        for (int j = 0; j < order; j++) o[j] = o[j] / scale[j];
    }
    return m;
}

/* ---------------------------------------------------------------------------------------
 * overscan.channel_solve (telescope ML1, accum='f32seq') in one call: the numpy glue between the
 * LAPACK fits, operation by operation as in blackbox_amd/overscan.py (channel_phase1,
 * vos_polyfit, channel_phase2, hos_fit), which follows blackbox.py:6497-6814.  The least-squares
 * solves themselves stay with numpy (np.linalg.lstsq = LAPACK gelsd, what np.polyfit calls):
 * [lstsq] is a callback into Python.  Situations the numpy code handles in other ways (failed or
 * rank-deficient vertical fit, non-finite fit, overscan pixels above data_limit, columns that
 * need the spline, too few points) return a positive code and the caller runs the numpy path.
 * tests/test_host_overscan.py holds both paths against each other bit for bit. */
typedef int (*bbx_lstsq_fn)(const double *lhs, int64_t m, int order, const double *rhs, double rcond, double *coef,
                            int *rank);

/* The same solve without the trip through the interpreter: LAPACK's dgelsd of the BLAS that numpy itself is
 * linked with (ILP64 build: scipy_dgelsd_64_), called the way numpy's umath_linalg does -- column-major copy of the
 * left-hand side, ldb = max(m, n), workspace sizes from a query call -- so that the same routine sees the same
 * numbers with the same block sizes.  The host installs the entry point (bbx_host_set_dgelsd); without it the
 * callback is used. */
typedef void (*bbx_dgelsd_fn)(int64_t *m, int64_t *n, int64_t *nrhs, double *a, int64_t *lda, double *b, int64_t *ldb,
                              double *s, double *rcond, int64_t *rank, double *work, int64_t *lwork, int64_t *iwork,
                              int64_t *info);
static bbx_dgelsd_fn g_dgelsd = 0;
void bbx_host_set_dgelsd(void *fn) { g_dgelsd = (bbx_dgelsd_fn)fn; }
int bbx_host_has_dgelsd(void) { return g_dgelsd != 0; }

int bbx_lstsq_direct(const double *lhs, int64_t m, int order, const double *rhs, double rcond, double *coef, int *rank) {
    if (!g_dgelsd || m < 1 || order < 1) return 1;
    int64_t mm = m, n = order, nrhs = 1, lda = m, ldb = m > n ? m : n, rk = 0, info = 0, lwork = -1, iwq = 0;
    double wq = 0.0, rc = rcond;
    /* singular values: min(m, n) entries, allocated with the matrices (no fixed-size buffer) */
    const size_t ns = (size_t)(m < n ? m : n);
    double *a = (double *)malloc(((size_t)m * order + (size_t)ldb + ns) * sizeof(double));
    if (!a) return 1;
    double *b = a + (size_t)m * order;
    double *sdummy = b + ldb;
    for (int j = 0; j < order; j++)
        for (int64_t i = 0; i < m; i++) a[(size_t)j * m + i] = lhs[(size_t)i * order + j];
    for (int64_t i = 0; i < ldb; i++) b[i] = i < m ? rhs[i] : 0.0;
    g_dgelsd(&mm, &n, &nrhs, a, &lda, b, &ldb, sdummy, &rc, &rk, &wq, &lwork, &iwq, &info);
    if (info != 0) { free(a); return 1; }
    lwork = (int64_t)wq;
    const int64_t liwork = iwq > 1 ? iwq : 1;
    double *work = (double *)malloc((size_t)(lwork > 1 ? lwork : 1) * sizeof(double) + (size_t)liwork * sizeof(int64_t));
    if (!work) { free(a); return 1; }
    int64_t *iwork = (int64_t *)(work + (lwork > 1 ? lwork : 1));
    g_dgelsd(&mm, &n, &nrhs, a, &lda, b, &ldb, sdummy, &rc, &rk, work, &lwork, iwork, &info);
    const int ok = info == 0;
    if (ok) { for (int j = 0; j < order; j++) coef[j] = b[j]; *rank = (int)rk; }
    free(work); free(a);
    return ok ? 0 : 1;
}

/* numpy's pairwise summation of a contiguous float64 vector (np.add.reduce: blocks of 128 with
 * eight accumulators, halves split at a multiple of 8) */
static double pairwise_sum(const double *a, int64_t n) {
    if (n < 8) {
        double r = 0.0;
        for (int64_t i = 0; i < n; i++) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; k++) r[k] = a[k];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

/* overscan._mean_std_flat(v, 'f64') */
static void mean_std_f64(const double *v, int64_t n, double *tmp, double *mean, double *sd) {
    const double m = pairwise_sum(v, n) / (double)n;
    for (int64_t i = 0; i < n; i++) { const double d = v[i] - m; tmp[i] = d * d; }
    *mean = m;
    *sd = sqrt(pairwise_sum(tmp, n) / (double)n);
}

/* np.polyval(p, start + arange(n)) with p high -> low order */
static void polyval_arange(const double *p, int order, int start, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++) {
        const double x = (double)(start + i);
        double y = 0.0;                                       /* zeros_like(x) (integers), then y * x + pv */
        for (int k = 0; k < order; k++) y = y * x + p[k];
        out[i] = y;
    }
}

/* polyfit_exact(start, n, mask, y[mask], order - 1) -> coefficients high -> low; 0 ok */
static int polyfit_masked(const double *V, const uint8_t *mask, int64_t n, int order, const double *yfull,
                          bbx_lstsq_fn lstsq, double *lhs, double *rhs, double *coef, int *rank) {
    double scale[16];
    const int64_t m = bbx_polyfit_prep(V, mask, n, order, lhs, scale);
    if (m < order) return 1;
    int64_t k = 0;
    for (int64_t i = 0; i < n; i++) if (mask[i]) rhs[k++] = yfull[i] + 0.0;
    const double rcond = (double)m * 2.220446049250313e-16;
    if ((lstsq ? lstsq(lhs, m, order, rhs, rcond, coef, rank) : bbx_lstsq_direct(lhs, m, order, rhs, rcond, coef, rank)) != 0) return 2;
    for (int j = 0; j < order; j++) coef[j] = coef[j] / scale[j];
    return 0;
}

#define IDX_SWITCH 150
#define OVERLAP 30
int bbx_channel_solve_ml1(int c, const double *mean_vos_col, int dy, const float *hos, int hos_rows, int dx, int ysz,
                          int xsz, int poldeg, double data_limit, const double *V_vos, const double *V_hos,
                          bbx_lstsq_fn lstsq, double *fit_out, double *coeffs_out, double *level_out,
                          double *dlevel_out, double *oscan_out) {
    const int order_v = poldeg + 1, order_h = 8;
    if (dy < 8 || xsz < 300 || xsz > dx || hos_rows < 1 || hos_rows > 4096 || order_v > 16) return 100;
    const size_t nmax = (size_t)(dy > xsz ? dy : xsz);
    static __thread void *scratch = 0;
    static __thread size_t scratch_bytes = 0;
    const size_t n_strip = (size_t)hos_rows * dx, n_blk = (size_t)hos_rows * (xsz + 300);
    const size_t need = (nmax * 21 + (size_t)xsz) * sizeof(double) + (n_strip + n_blk + 3 * (size_t)xsz + 4) * sizeof(float)
                        + 2 * nmax + (size_t)hos_rows * xsz + 64;
    if (scratch_bytes < need) {
        free(scratch);
        scratch = malloc(need);
        scratch_bytes = scratch ? need : 0;
        if (!scratch) return -3;
    }
    double *v = (double *)scratch, *tmp = v + nmax, *rhs = tmp + nmax, *lhs = rhs + nmax;   /* lhs: nmax * 16 */
    double *yd = lhs + nmax * 16;                                                            /* nmax */
    int64_t *ncol = (int64_t *)(yd + nmax);                                                  /* xsz */
    float *strip = (float *)(ncol + xsz);                                                    /* hos_rows * dx */
    float *blk = strip + n_strip;                                                            /* hos_rows * (xsz + 300) */
    float *mean_hos = blk + n_blk, *std_hos = mean_hos + xsz, *err_hos = std_hos + xsz;
    uint8_t *mask = (uint8_t *)(err_hos + xsz + 4), *mask2 = mask + nmax, *zero_mask = mask2 + nmax;

    /* ---- vos_polyfit ---------------------------------------------------------------- */
    int64_t m = 0;
    for (int i = 0; i < dy; i++) if (isfinite(mean_vos_col[i])) v[m++] = mean_vos_col[i];
    double mean = NAN, sd = NAN;
    for (int it = 0; it < 5; it++) {
        if (m == 0) break;
        mean_std_f64(v, m, tmp, &mean, &sd);
        const double lo = mean - sd * 5.0, hi = mean + sd * 5.0;
        int64_t k = 0;
        for (int64_t i = 0; i < m; i++) if (v[i] >= lo && v[i] <= hi) v[k++] = v[i];
        if (k == m) break;
        m = k;
    }
    if (m == 0) return 1;                                     /* statistics undefined: numpy path */
    mean_std_f64(v, m, tmp, &mean, &sd);
    for (int i = 0; i < dy; i++)
        mask[i] = (sd == 0.0) ? 1 : (uint8_t)(fabs(mean_vos_col[i] - mean) / sd <= 5.0);
    if (c < 8) { for (int i = ysz; i < dy; i++) mask[i] = 0; }
    else { for (int i = 0; i < dy - ysz; i++) mask[i] = 0; }
    double p[16];
    int rank = 0;
    if (polyfit_masked(V_vos, mask, dy, order_v, mean_vos_col, lstsq, lhs, rhs, p, &rank) != 0) return 2;
    if (rank != order_v) return 3;
    polyval_arange(p, order_v, 0, dy, fit_out);
    for (int i = 0; i < dy; i++) if (!isfinite(fit_out[i])) return 4;
    *level_out = pairwise_sum(fit_out, dy) / (double)dy;      /* np.mean(fit) */
    for (int k = 0; k < order_v; k++) coeffs_out[k] = p[order_v - 1 - k];

    /* ---- horizontal strip after the vertical fit; its level ------------------------ */
    const int rl0 = (c < 8) ? (dy - hos_rows) : 0;
    for (int r = 0; r < hos_rows; r++) {
        const double f = fit_out[rl0 + r];
        for (int x = 0; x < dx; x++) strip[(size_t)r * dx + x] = (float)((double)hos[(size_t)r * dx + x] - f);
    }
    float *lvl = blk;
    for (int r = 0; r < hos_rows; r++)
        for (int x = 0; x < 300; x++) lvl[(size_t)r * 300 + x] = strip[(size_t)r * dx + (xsz - 300) + x];
    double st[2];
    const int64_t ml = bbx_clipped_stats_flat_f32seq(lvl, (int64_t)hos_rows * 300, 3.0, 5, st);
    if (ml <= 0) return 5;
    const float dlevel = (float)st[0];
    *dlevel_out = (double)dlevel;
    float *data_hos = blk;                                    /* [hos_rows][xsz], contiguous */
    for (int r = 0; r < hos_rows; r++)
        for (int x = 0; x < xsz; x++) {
            const float s = strip[(size_t)r * dx + x] - dlevel;
            if (s > (float)data_limit) return 6;              /* hos_mask_ml1 would mask something */
            data_hos[(size_t)r * xsz + x] = s;
        }
    memset(zero_mask, 0, (size_t)hos_rows * xsz);
    if (bbx_hos_column_stats_f32seq(data_hos, zero_mask, hos_rows, xsz, ncol, mean_hos, std_hos) != 0) return -3;

    /* ---- hos_fit --------------------------------------------------------------------- */
    for (int x = 0; x < xsz; x++) {
        err_hos[x] = 0.0f;
        if (ncol[x] > 1) err_hos[x] = (float)((double)std_hos[x] / sqrt((double)ncol[x]));
    }
    /* columns below IDX_SWITCH that do not take their own mean need the spline: numpy path */
    for (int x = 0; x < IDX_SWITCH; x++) {
        const int valid = ncol[x] > 1;
        int keeps_spline = !valid;
        if (x < 3) keeps_spline = keeps_spline && !valid;
        if (keeps_spline) return 7;
    }
    int64_t nm = 0;
    for (int x = 0; x < xsz; x++) {
        mask[x] = (uint8_t)(ncol[x] > 1 && x >= IDX_SWITCH - OVERLAP);
        if (mask[x]) blk[nm++] = mean_hos[x];                 /* mhp (data_hos is not needed any more) */
    }
    const int64_t mk = bbx_clipped_stats_flat_f32seq(blk, nm, 5.0, 5, st);
    if (mk <= 0) return 8;
    const float cm = (float)st[0], cs = (float)st[1];
    if (!(cs == 0.0f)) {
        for (int x = 0; x < xsz; x++)
            if (mask[x]) mask[x] = (uint8_t)(fabsf(mean_hos[x] - cm) / cs <= 5.0f);
    }
    for (int x = 0; x < xsz; x++) yd[x] = (double)mean_hos[x];
    double ph[16];
    for (int it = 0; it < 3; it++) {
        int rk;
        if (polyfit_masked(V_hos, mask, xsz, order_h, yd, lstsq, lhs, rhs, ph, &rk) != 0) return 9;
        polyval_arange(ph, order_h, 1, xsz, oscan_out);
        int same = 1;
        for (int x = 0; x < xsz; x++) {
            const double e3 = (double)(3.0f * err_hos[x]);
            const uint8_t nmk = (uint8_t)(mask[x] && (fabs(oscan_out[x] - yd[x]) <= e3));
            if (nmk != mask[x]) same = 0;
            mask2[x] = nmk;
        }
        if (same) break;
        memcpy(mask, mask2, (size_t)xsz);
    }
    for (int x = 0; x < IDX_SWITCH; x++)                       /* mask_usemean (and the first three columns) */
        if (ncol[x] > 1) oscan_out[x] = (double)mean_hos[x];
    return 0;
}
