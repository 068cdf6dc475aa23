/* bbx_host.c -- host-side helpers of the overscan solve (plain C, no GPU): the per-column
 * sigma-clipped statistics of the horizontal-overscan strip and the flat clipped
 * statistics, written out with exactly the float operations (and their order) of the
 * numpy code in blackbox_amd/overscan.py, which in turn follows the reference
 * (blackbox.py:6649-6659, astropy sigma_clip).  Build: gcc -O2 -ffp-contract=off.
 * tests/test_host_overscan.py holds both against the numpy versions bit for bit. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* overscan.hos_column_stats(data_hos, mask_hos, accum='f32seq'):
 * data [nrow][ncol] float32, mask [nrow][ncol] uint8 (non-zero = masked)
 * -> n [ncol] int64, mean [ncol] float32, std [ncol] float32 */
int bbx_hos_column_stats_f32seq(const float *data, const uint8_t *mask, int nrow, int ncol, int64_t *n_out,
                                float *mean_out, float *std_out) {
    if (nrow < 1 || nrow > 4096) return -1;
    uint8_t *ok = (uint8_t *)malloc((size_t)nrow), *cur = (uint8_t *)malloc((size_t)nrow);
    if (!ok || !cur) { free(ok); free(cur); return -3; }
    for (int x = 0; x < ncol; x++) {
        double lo = NAN, hi = NAN;
        for (int i = 0; i < nrow; i++) {
            const double d = (double)data[(size_t)i * ncol + x];
            ok[i] = (uint8_t)(isfinite(d) && !mask[(size_t)i * ncol + x]);
            cur[i] = ok[i];
        }
        for (int it = 0; it < 5; it++) {
            int64_t n = 0;
            double s = 0.0;
            for (int i = 0; i < nrow; i++) {
                const double d = (double)data[(size_t)i * ncol + x];
                const double t = cur[i] ? d : 0.0;
                s = (i == 0) ? t : s + t;                      /* add.reduce over axis 0: row 0 first */
                n += cur[i];
            }
            const double mean = s / (double)n;
            double q = 0.0;
            for (int i = 0; i < nrow; i++) {
                const double d = (double)data[(size_t)i * ncol + x];
                const double dev = cur[i] ? (mean - d) : 0.0;
                const double t = dev * dev;
                q = (i == 0) ? t : q + t;
            }
            const double sd = sqrt(q / (double)n);
            if (n > 0) { lo = mean - 2.5 * sd; hi = mean + 2.5 * sd; }
            for (int i = 0; i < nrow; i++) {
                const double d = (double)data[(size_t)i * ncol + x];
                cur[i] = (uint8_t)(cur[i] && d >= lo && d <= hi);
            }
        }
        int64_t n = 0;
        for (int i = 0; i < nrow; i++) {
            const double d = (double)data[(size_t)i * ncol + x];
            ok[i] = (uint8_t)(ok[i] && !(d < lo) && !(d > hi));
            n += ok[i];
        }
        float tot = 0.0f;
        for (int i = 0; i < nrow; i++) tot = tot + (ok[i] ? data[(size_t)i * ncol + x] : 0.0f);
        const float mean = tot / (float)n;
        float tot2 = 0.0f;
        for (int i = 0; i < nrow; i++) {
            const float dev = ok[i] ? (data[(size_t)i * ncol + x] - mean) : 0.0f;
            tot2 = tot2 + dev * dev;
        }
        n_out[x] = n;
        mean_out[x] = mean;
        std_out[x] = sqrtf(tot2 / (float)(n - 1));
    }
    free(ok); free(cur);
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
