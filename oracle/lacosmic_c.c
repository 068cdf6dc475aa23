/* TEST INFRASTRUCTURE -- the C twin of oracle/lacosmic.py (same algorithm, same float32 operation order, same
 * conventions: see that file's header), for CPU timing at sizes the numpy version is too slow for
 * (bench.py's cpu_baseline) and as a second, independent statement of the algorithm.
 *
 * PARITY UNPINNED like lacosmic.py: astroscrappy (C / OpenMP in the reference's environment, blackbox.py:4323-4332,
 * OMP_NUM_THREADS 34-39) is not in /root/reference; tests/test_lacosmic_oracle.py holds this file bit for bit against
 * lacosmic.py.  Only tests/, bench.py's cpu_baseline leg and __graft_entry__ may load it; nothing under blackbox_amd/ does.
 *
 * build: gcc -O3 -fopenmp -ffp-contract=off -fPIC -shared -o oracle/liblacosmic_c.so oracle/lacosmic_c.c -lm   (Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* k-th smallest of v[0..n) (v is reordered) */
static float select_k(float *v, long n, long k) {
    long lo = 0, hi = n - 1;
    while (lo < hi) {
        const float p = v[(lo + hi) / 2];
        long i = lo, j = hi;
        while (i <= j) {
            while (v[i] < p) i++;
            while (v[j] > p) j--;
            if (i <= j) { const float t = v[i]; v[i] = v[j]; v[j] = t; i++; j--; }
        }
        if (k <= j) hi = j; else if (k >= i) lo = i; else return v[k];
    }
    return v[k];
}

/* K x K median (odd K), the outer K/2 rows / columns copied from the input.  The median is an order statistic: any
 * correct selection gives the same bits.  Here: Batcher's odd-even merge sorting network on the K*K window values of XB
 * neighbouring pixels at once (one "wire" = XB floats, a compare-exchange = a vector min + max; the window padded to a
 * power of two with +inf wires, whose comparators are no-ops and are dropped) -- astroscrappy uses fixed networks for its
 * 3 x 3 / 5 x 5 medians as well (PyOptMed9 / PyOptMed25). */
#define XB 16
typedef struct { int n, ncmp; unsigned char (*cmp)[2]; } sortnet;
static sortnet nets[3];

static const sortnet *get_net(int nwin) {
    sortnet *net = &nets[nwin == 9 ? 0 : (nwin == 25 ? 1 : 2)];
    if (net->cmp) return net;
    sortnet t; t.n = nwin; t.ncmp = 0;
    int P = 1; while (P < nwin) P *= 2;
    t.cmp = malloc(sizeof(unsigned char[2]) * 2048);
    for (int p = 1; p < P; p *= 2)
        for (int k = p; k >= 1; k /= 2)
            for (int j = k % p; j <= P - 1 - k; j += 2 * k)
                for (int i = 0; i <= (k - 1 < P - j - k - 1 ? k - 1 : P - j - k - 1); i++)
                    if ((i + j) / (2 * p) == (i + j + k) / (2 * p) && i + j + k < nwin) {      /* (a +inf wire never changes) */
                        t.cmp[t.ncmp][0] = (unsigned char)(i + j); t.cmp[t.ncmp][1] = (unsigned char)(i + j + k); t.ncmp++;
                    }
#pragma omp critical
    { if (!net->cmp) *net = t; else free(t.cmp); }
    return net;
}

__attribute__((target_clones("avx2", "default")))
static void medfilt_rows(const float *a, float *out, int nx, int K, int y, const sortnet *net) {
    const int h = K / 2, n = K * K;
    float w[49][XB];
    for (int x0 = h; x0 < nx - h; x0 += XB) {
        const int nv = (nx - h - x0 < XB) ? nx - h - x0 : XB;                  /* valid lanes of this group */
        int m = 0;
        for (int dy = -h; dy <= h; dy++)
            for (int dx = -h; dx <= h; dx++, m++) {
                const float *src = a + (size_t)(y + dy) * nx + x0 + dx;
                if (nv == XB) for (int l = 0; l < XB; l++) w[m][l] = src[l];
                else for (int l = 0; l < XB; l++) w[m][l] = src[l < nv ? l : nv - 1];
            }
        for (int c = 0; c < net->ncmp; c++) {
            float *restrict p = w[net->cmp[c][0]], *restrict q = w[net->cmp[c][1]];       /* (two different wires) */
#pragma omp simd
            for (int l = 0; l < XB; l++) { const float u = p[l], v = q[l]; p[l] = u < v ? u : v; q[l] = u < v ? v : u; }
        }
        for (int l = 0; l < nv; l++) out[(size_t)y * nx + x0 + l] = w[n / 2][l];
    }
}

static void medfilt(const float *a, float *out, int ny, int nx, int K) {
    const int h = K / 2;
    memcpy(out, a, (size_t)ny * nx * sizeof(float));
    if (ny <= 2 * h || nx <= 2 * h) return;
    const sortnet *net = get_net(K * K);
#pragma omp parallel for schedule(static)
    for (int y = h; y < ny - h; y++) medfilt_rows(a, out, nx, K, y, net);
}

/* 3 x 3 binary dilation, the 1-pixel border copied */
static void dilate3(const uint8_t *m, uint8_t *out, int ny, int nx) {
    memcpy(out, m, (size_t)ny * nx);
    if (ny <= 2 || nx <= 2) return;
#pragma omp parallel for schedule(static)
    for (int y = 1; y < ny - 1; y++)
        for (int x = 1; x < nx - 1; x++) {
            uint8_t v = 0;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) v |= m[(size_t)(y + dy) * nx + x + dx];
            out[(size_t)y * nx + x] = v;
        }
}

/* L+: 2 x 2 replication, Laplacian with dropped outside neighbours (order: 4c, -right, -left, -down, -up), clip at 0,
 * 2 x 2 mean ((tl + tr) + bl) + br, * 0.25 */
static void lplus(const float *c, float *out, int ny, int nx) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < ny; y++)
        for (int x = 0; x < nx; x++) {
            float sub[4];
            const float v = c[(size_t)y * nx + x];
            for (int sy = 0; sy < 2; sy++)
                for (int sx = 0; sx < 2; sx++) {
                    const int Y = 2 * y + sy, X = 2 * x + sx;                 /* sub-pixel coordinates in the replicated image */
                    float p = 4.0f * v;
                    if (X + 1 < 2 * nx) p -= c[(size_t)y * nx + (X + 1) / 2];
                    if (X - 1 >= 0) p -= c[(size_t)y * nx + (X - 1) / 2];
                    if (Y + 1 < 2 * ny) p -= c[(size_t)((Y + 1) / 2) * nx + x];
                    if (Y - 1 >= 0) p -= c[(size_t)((Y - 1) / 2) * nx + x];
                    sub[2 * sy + sx] = p > 0.0f ? p : 0.0f;
                }
            out[(size_t)y * nx + x] = (((sub[0] + sub[1]) + sub[2]) + sub[3]) * 0.25f;
        }
}

/* astroscrappy.detect_cosmics(sepmed=False, cleantype='medmask', fsmode='median', gain=1, pssl=0, satlevel=inf) as the
 * reference calls it; inmask: non-zero = masked.  crmask (uint8 0/1) and clean are outputs; n_iter[k] = pixels flagged in
 * iteration k.  -> 0, or -1 when memory runs out */
int lac_detect_cosmics(const float *indat, const uint8_t *inmask, int ny, int nx, float sigclip, float sigfrac, float objlim,
                       int niter, float readnoise, uint8_t *crmask, float *clean, long *n_iter, int nthreads) {
    const size_t N = (size_t)ny * nx;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    float *lp = malloc(N * 4), *m5 = malloc(N * 4), *s = malloc(N * 4), *ms = malloc(N * 4), *m3 = malloc(N * 4), *m37 = malloc(N * 4);
    uint8_t *cos = malloc(N), *dil = malloc(N);
    if (!lp || !m5 || !s || !ms || !m3 || !m37 || !cos || !dil) { free(lp); free(m5); free(s); free(ms); free(m3); free(m37); free(cos); free(dil); return -1; }
    memcpy(clean, indat, N * 4);
    memset(crmask, 0, N);
    /* background level: lower median of the unmasked input pixels */
    float background = 0.0f;
    {
        long ng = 0;
        for (size_t i = 0; i < N; i++) if (!inmask[i]) lp[ng++] = clean[i];
        if (ng) background = select_k(lp, ng, (ng - 1) / 2);
    }
    const float sigcliplow = sigfrac * sigclip;
    const float rn2 = readnoise * readnoise;
    for (int it = 0; it < niter; it++) {
        lplus(clean, lp, ny, nx);
        medfilt(clean, m5, ny, nx, 5);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < N; i++) {
            const float m = m5[i] > 0.00001f ? m5[i] : 0.00001f;
            const float noise = sqrtf(m + rn2);
            m5[i] = noise;                                         /* m5 holds the noise from here on */
            s[i] = lp[i] / (2.0f * noise);
        }
        medfilt(s, ms, ny, nx, 5);
        medfilt(clean, m3, ny, nx, 3);
        medfilt(m3, m37, ny, nx, 7);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < N; i++) {
            const float sp = s[i] - ms[i];
            float f = (m3[i] - m37[i]) / m5[i];
            if (f < 0.01f) f = 0.01f;
            s[i] = sp;                                             /* s holds sp from here on */
            cos[i] = (sp > sigclip) && !inmask[i] && ((sp / f) > objlim);
        }
        dilate3(cos, dil, ny, nx);
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < N; i++) cos[i] = dil[i] && !inmask[i] && (s[i] > sigclip);
        dilate3(cos, dil, ny, nx);
        long n = 0;
#pragma omp parallel for schedule(static) reduction(+ : n)
        for (size_t i = 0; i < N; i++) {
            const uint8_t c = dil[i] && !inmask[i] && (s[i] > sigcliplow);
            cos[i] = c;
            n += c;
            crmask[i] |= c;
        }
        if (n_iter) n_iter[it] = n;
        if (n == 0) break;
        /* clean_medmask: every CR pixel (2 pixels off the border) <- lower median of the unflagged, unmasked 5 x 5 neighbours */
        memcpy(lp, clean, N * 4);                                  /* windows never read flagged pixels: a snapshot keeps the loop parallel */
#pragma omp parallel for schedule(dynamic, 16)
        for (int y = 2; y < ny - 2; y++)
            for (int x = 2; x < nx - 2; x++) {
                if (!crmask[(size_t)y * nx + x]) continue;
                float w[25];
                int m = 0;
                for (int dy = -2; dy <= 2; dy++)
                    for (int dx = -2; dx <= 2; dx++) {
                        const size_t k = (size_t)(y + dy) * nx + x + dx;
                        if (!crmask[k] && !inmask[k]) w[m++] = lp[k];
                    }
                clean[(size_t)y * nx + x] = m ? select_k(w, m, (m - 1) / 2) : background;
            }
    }
    free(lp); free(m5); free(s); free(ms); free(m3); free(m37); free(cos); free(dil);
    return 0;
}
