// bbx_sat.hip -- satellite-trail masking (reference sat_detect, blackbox.py:4163-4254).
// The reference calls acstools.satdet (or ASTA, a CNN without weights here); acstools is absent from this
// image.  The detector follows acstools' published chain where it is deterministic and is specified in
// oracle/sattrail.py:
//
//   2x2 sum binning -> [bbx_canny.hip: percentile rescale, Canny(sigma 3, 0.1 / 0.2 of the maximum),
//   remove_small_objects(60): pinned against scikit-image] -> full Hough accumulator over acstools' theta grid
//   (2 .. 177.5 deg, skimage.transform.hough_line's cells; stands in for the *random* probabilistic transform)
//   -> best cell with >= 210 votes whose supporting edge pixels span the frame to within buf = 40 px of both
//   borders -> segment = longest run of that cell's voters (the deterministic stand-in for a segment of the random
//   transform) -> acstools.satdet.make_mask(segment, sublen = 5, pad = 0, sigma = 5) as restated in
//   oracle/sattrail.py: image / max rotated so that the trail runs along the rows (skimage.transform.rotate, order 3:
//   only the band of rows the walk can reach is ever interpolated), windows of 200 columns x 10 rows along the trail
//   (row medians, clipped mean + sigma * biweight midvariance, the window centre follows the trail), the strips
//   rotated back (order 1) -> bit 16.  Every float64 operation is the one the oracle does, in its order.
// Traffic: one read of the frame (4N) for binning, then everything works on the 4x smaller binned frame and on
// sparse pixel lists.
#include "bbx_common.h"

#define SAT_BLOCKS 1024

struct sat_state {
    unsigned long long best;              // (votes << 32) | (0xffffffff - flat index)
    int accept, found, k, lo_off, hi_off; // line accepted / strip found, theta index, strip offsets
    int votes;
    double rho, c, s, chord;
    unsigned long long tmin, tmax;        // ordered keys of the smallest / largest position of a supporting edge pixel along the line
    double t0, t1;                        // where the line enters / leaves the frame
    unsigned long long fit_n; long long fit_t, fit_d, fit_tt, fit_td;   // integer sums (1/16 px) of the refinement
    double icpt, slope;                   // refined line: distance d - (icpt + slope t)
    // ---- make_mask
    unsigned int bmax_key, bmin_key;      // ordered keys of the largest / smallest binned pixel
    int seg[4], seg_ok;                   // x0, y0, x1, y1 of the segment handed to make_mask
    double m[6], m2[6];                   // rotation there (output -> input) and back
    int rot_rows, rot_cols, back_rows, back_cols, crop_y, crop_x;
    double sx, sy;                        // the segment's first point in the rotated frame
    double top, clip_lo, clip_hi;         // normalisation, clip range of the interpolation
    int band_r0, band_rows;               // rows of the rotated frame that are interpolated
    int nwin, row_min, row_max;           // strips painted (rotated frame), their row range
};
#define SAT_BANDH 96                      // the walk may drift this many rows from its start
#define SAT_MAXWIN 1200
#define SAT_SEG_CAP 8192
#define SAT_SUBLEN 5
#define SAT_SUBW 200
struct sat_rect { int r0, r1, c0, c1; };

__device__ __forceinline__ unsigned fkey(float f) { const unsigned u = __float_as_uint(f); return (u >> 31) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float fkey_inv(unsigned k) { return __uint_as_float((k >> 31) ? (k & 0x7fffffffu) : ~k); }

// 2 x 2 sums (blackbox.py:4171-4172)
__global__ __launch_bounds__(256) void k_bin2(const float* __restrict__ d, int nyb, int nxb, float* __restrict__ b) {
    const int X = blockIdx.x * blockDim.x + threadIdx.x, Y = blockIdx.y;
    if (X >= nxb) return;
    const size_t nx = (size_t)nxb * 2;
    const float2 r0 = *(const float2*)(d + (size_t)(2 * Y) * nx + 2 * X);
    const float2 r1 = *(const float2*)(d + (size_t)(2 * Y + 1) * nx + 2 * X);
    b[(size_t)Y * nxb + X] = (r0.x + r0.y) + (r1.x + r1.y);
}

// the largest and the smallest binned value (make_mask divides by the maximum; the interpolation is clipped to the image's
// range): 16-byte loads, one pair of atomics per workgroup (same-address atomics retire one at a time, ~11 ns each)
__global__ __launch_bounds__(256) void k_bin_minmax(const float* __restrict__ b, size_t n, sat_state* st) {
    unsigned hi = 0u, lo = ~0u;
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * blockDim.x;
    const float4* b4 = (const float4*)b;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 q = b4[i];
        const float f[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int u = 0; u < 4; u++) if (isfinite(f[u])) { const unsigned k = fkey(f[u]); hi = k > hi ? k : hi; lo = k < lo ? k : lo; }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float f = b[(n4 << 2) + threadIdx.x]; if (isfinite(f)) { const unsigned k = fkey(f); hi = k > hi ? k : hi; lo = k < lo ? k : lo; } }
    for (int o = 32; o > 0; o >>= 1) { const unsigned a = __shfl_xor(hi, o, 64), c = __shfl_xor(lo, o, 64); hi = a > hi ? a : hi; lo = c < lo ? c : lo; }
    __shared__ unsigned shi[4], slo[4];
    if ((threadIdx.x & 63) == 0) { shi[threadIdx.x >> 6] = hi; slo[threadIdx.x >> 6] = lo; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) { hi = shi[w] > hi ? shi[w] : hi; lo = slo[w] < lo ? slo[w] : lo; }
        if (hi) { atomicMax(&st->bmax_key, hi); atomicMin(&st->bmin_key, lo); }
    }
}

__global__ void k_sat_init(sat_state* st) {
    if (threadIdx.x == 0) {
        st->best = 0;
        st->accept = 0; st->found = 0; st->k = 0; st->lo_off = 0; st->hi_off = 0; st->votes = 0;
        st->rho = 0; st->c = 0; st->s = 0; st->chord = 0; st->tmin = ~0ull; st->tmax = 0ull; st->t0 = 0; st->t1 = 0;
        st->fit_n = 0; st->fit_t = st->fit_d = st->fit_tt = st->fit_td = 0; st->icpt = 0; st->slope = 0;
        st->bmax_key = 0u; st->bmin_key = ~0u; st->seg_ok = 0; st->nwin = 0; st->row_min = 0x7fffffff; st->row_max = -1;
        st->band_r0 = 0; st->band_rows = 0; st->rot_rows = st->rot_cols = 0;
    }
}

// Hough votes: one block per angle keeps the rho histogram of that angle in LDS and reports
// only its best cell -- no global accumulator.  Order of the edge list does not matter.
__global__ __launch_bounds__(1024) void k_hough_lds(const uint32_t* __restrict__ list, const int32_t* __restrict__ cnt, uint32_t cap,
                                                    int nxb, const double* __restrict__ cs, int nrho, sat_state* st) {
    extern __shared__ unsigned hist[];
    const int k = blockIdx.x;
    const uint32_t n = min((uint32_t)*cnt, cap);
    for (int i = threadIdx.x; i < nrho; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const double c = cs[2 * k], s = cs[2 * k + 1];
    const int off = nrho / 2;
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
        const uint32_t p = list[e];
        const double y = (double)(p / nxb), x = (double)(p % nxb);
        const double r = c * x + s * y;
        atomicAdd(&hist[(int)(r > 0.0 ? r + 0.5 : r - 0.5) + off], 1u);       // skimage.transform.hough_line: round half away from zero
    }
    __syncthreads();
    unsigned long long best = 0;
    for (int i = threadIdx.x; i < nrho; i += blockDim.x) {
        const unsigned v = hist[i];
        const unsigned long long flat = (unsigned long long)k * nrho + i;
        const unsigned long long key = ((unsigned long long)v << 32) | (0xffffffffull - flat);
        if (v && key > best) best = key;
    }
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(best, o, 64); if (t > best) best = t; }
    if ((threadIdx.x & 63) == 0 && best) atomicMax(&st->best, best);
}

__device__ double chord_length(double c, double s, double rho, int ny, int nx) {
    double px[4], py[4]; int m = 0;
    const double xs[2] = {0.0, nx - 1.0}, ys[2] = {0.0, ny - 1.0};
    for (int i = 0; i < 2; i++) if (fabs(s) > 1e-12) { const double y = (rho - xs[i] * c) / s; if (y >= -1e-9 && y <= ny - 1 + 1e-9) { px[m] = xs[i]; py[m] = y; m++; } }
    for (int i = 0; i < 2; i++) if (fabs(c) > 1e-12) { const double x = (rho - ys[i] * s) / c; if (x >= -1e-9 && x <= nx - 1 + 1e-9) { px[m] = x; py[m] = ys[i]; m++; } }
    double best = 0.0;
    for (int i = 0; i < m; i++) for (int j = i + 1; j < m; j++) { const double d = hypot(px[i] - px[j], py[i] - py[j]); if (d > best) best = d; }
    return m < 2 ? 0.0 : best;
}

__device__ __forceinline__ unsigned long long dkey(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dkey_inv(unsigned long long k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7fffffffffffffffull) : ~k));
}

__global__ void k_trail_decide(sat_state* st, const double* __restrict__ cs, int nrho, int nyb, int nxb, int min_votes) {
    if (threadIdx.x != 0) return;
    const unsigned long long b = st->best;
    if (!b) return;
    const unsigned votes = (unsigned)(b >> 32);
    const unsigned long long flat = 0xffffffffull - (b & 0xffffffffull);
    const int k = (int)(flat / nrho), r = (int)(flat % nrho);
    st->k = k; st->votes = (int)votes; st->rho = (double)(r - nrho / 2); st->c = cs[2 * k]; st->s = cs[2 * k + 1];
    st->chord = chord_length(st->c, st->s, st->rho, nyb, nxb);
    st->accept = (votes >= (unsigned)min_votes) ? 1 : 0;          // provisional: the support test follows
    // where the line x c + y s = rho crosses the frame rectangle, as positions t = y c - x s along it
    const double c = st->c, s = st->s, rho = st->rho;
    double t0 = __builtin_huge_val(), t1 = -__builtin_huge_val(); int m = 0;
    const double xs[2] = {0.0, nxb - 1.0}, ys[2] = {0.0, nyb - 1.0};
    for (int i = 0; i < 2; i++) if (fabs(s) > 1e-12) { const double y = (rho - xs[i] * c) / s; if (y >= -1e-9 && y <= nyb - 1 + 1e-9) { const double t = y * c - xs[i] * s; t0 = fmin(t0, t); t1 = fmax(t1, t); m++; } }
    for (int i = 0; i < 2; i++) if (fabs(c) > 1e-12) { const double x = (rho - ys[i] * s) / c; if (x >= -1e-9 && x <= nxb - 1 + 1e-9) { const double t = ys[i] * c - x * s; t0 = fmin(t0, t); t1 = fmax(t1, t); m++; } }
    st->t0 = t0; st->t1 = t1;
    if (m < 2) st->accept = 0;
}

// The best cell fixes the line to half a grid step (0.25 deg): refine it on the edge pixels inside a cone of that
// opening -- least squares d = a + b t on 1/16-px fixed-point coordinates with integer sums (order-independent)
#define SAT_TAN_HALF_STEP 0.004363350820701567
// round 0: the cone around the cell's line; later rounds (tol > 0): the pixels within tol of the line fitted so far
__global__ __launch_bounds__(256) void k_trail_cone(const uint32_t* __restrict__ list, const int32_t* __restrict__ cnt, uint32_t cap, int nxb,
                                                    sat_state* st, double tol) {
    if (!st->accept) return;
    const uint32_t n = min((uint32_t)*cnt, cap);
    const double c = st->c, s = st->s, rho = st->rho, tm = 0.5 * (st->t0 + st->t1), icpt = st->icpt, slope = st->slope;
    long long a_n = 0, a_t = 0, a_d = 0, a_tt = 0, a_td = 0;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const uint32_t p = list[e];
        const double y = (double)(p / nxb), x = (double)(p % nxb);
        const double d = c * x + s * y - rho, t = y * c - x * s;
        if (tol > 0.0 ? fabs(d - (icpt + slope * t)) <= tol : fabs(d) <= 1.5 + SAT_TAN_HALF_STEP * fabs(t - tm)) {
            const long long tq = (long long)floor(t * 16.0 + 0.5), dq = (long long)floor(d * 16.0 + 0.5);
            a_n++; a_t += tq; a_d += dq; a_tt += tq * tq; a_td += tq * dq;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        a_n += __shfl_xor(a_n, o, 64); a_t += __shfl_xor(a_t, o, 64); a_d += __shfl_xor(a_d, o, 64);
        a_tt += __shfl_xor(a_tt, o, 64); a_td += __shfl_xor(a_td, o, 64);
    }
    if ((threadIdx.x & 63) == 0 && a_n) {
        atomicAdd(&st->fit_n, (unsigned long long)a_n);
        atomicAdd((unsigned long long*)&st->fit_t, (unsigned long long)a_t); atomicAdd((unsigned long long*)&st->fit_d, (unsigned long long)a_d);
        atomicAdd((unsigned long long*)&st->fit_tt, (unsigned long long)a_tt); atomicAdd((unsigned long long*)&st->fit_td, (unsigned long long)a_td);
    }
}

__global__ void k_trail_refine(sat_state* st) {
    if (threadIdx.x != 0 || !st->accept) return;
    if (st->fit_n < 2) { st->accept = 0; return; }
    const double n = (double)st->fit_n;
    const double mt = (double)st->fit_t / n, md = (double)st->fit_d / n;
    const double var = (double)st->fit_tt / n - mt * mt, cov = (double)st->fit_td / n - mt * md;
    const double slope = var > 0.0 ? cov / var : 0.0;
    st->slope = slope;
    st->icpt = (md - slope * mt) / 16.0;
    st->fit_n = 0; st->fit_t = st->fit_d = st->fit_tt = st->fit_td = 0;      // for the next round
}

// extent along the line of the edge pixels within 1.5 px of the refined line
__global__ __launch_bounds__(256) void k_trail_support(const uint32_t* __restrict__ list, const int32_t* __restrict__ cnt, uint32_t cap, int nxb,
                                                       sat_state* st) {
    if (!st->accept) return;
    const uint32_t n = min((uint32_t)*cnt, cap);
    const double c = st->c, s = st->s, rho = st->rho, icpt = st->icpt, slope = st->slope;
    unsigned long long lo = ~0ull, hi = 0ull;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const uint32_t p = list[e];
        const double y = (double)(p / nxb), x = (double)(p % nxb);
        const double d = c * x + s * y - rho, t = y * c - x * s;
        if (fabs(d - (icpt + slope * t)) <= 1.5) { const unsigned long long k = dkey(t); lo = k < lo ? k : lo; hi = k > hi ? k : hi; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long a = __shfl_xor(lo, o, 64), b = __shfl_xor(hi, o, 64);
        lo = a < lo ? a : lo; hi = b > hi ? b : hi;
    }
    if ((threadIdx.x & 63) == 0 && hi) { atomicMin(&st->tmin, lo); atomicMax(&st->tmax, hi); }
}

__global__ void k_trail_accept(sat_state* st, double buf) {
    if (threadIdx.x != 0 || !st->accept) return;
    if (!st->tmax) { st->accept = 0; return; }
    const double tmin = dkey_inv(st->tmin), tmax = dkey_inv(st->tmax);
    st->accept = (tmin - st->t0 <= buf && st->t1 - tmax <= buf) ? 1 : 0;
}

// ---- make_mask ---------------------------------------------------------------------------------------------------
// rotate_matrix of oracle/sattrail.py: one IEEE operation per written operation, in that order
__device__ void sat_rotate_matrix(int rows, int cols, double dirx, double diry, double* m, int* out_rows, int* out_cols) {
    const double h = sqrt(dirx * dirx + diry * diry);
    const double C = dirx / h, S = diry / h;
    const double cx = cols / 2.0 - 0.5, cy = rows / 2.0 - 0.5;
    const double tx = (C * (-cx) + (-S) * (-cy)) + cx;
    const double ty = (S * (-cx) + C * (-cy)) + cy;
    const double px[4] = {0.0, 0.0, cols - 1.0, cols - 1.0}, py[4] = {0.0, rows - 1.0, rows - 1.0, 0.0};
    double minc = __builtin_huge_val(), maxc = -__builtin_huge_val(), minr = minc, maxr = maxc;
    for (int i = 0; i < 4; i++) {
        const double ax = px[i] - tx, ay = py[i] - ty;
        const double u = C * ax + S * ay, v = (-S) * ax + C * ay;
        minc = fmin(minc, u); maxc = fmax(maxc, u); minr = fmin(minr, v); maxr = fmax(maxr, v);
    }
    *out_rows = (int)rint(maxr - minr + 1); *out_cols = (int)rint(maxc - minc + 1);
    m[0] = C; m[1] = -S; m[2] = (C * minc + (-S) * minr) + tx;
    m[3] = S; m[4] = C; m[5] = (S * minc + C * minr) + ty;
}

// The segment: voters of the best cell ordered along the line, longest stretch without a gap > 75 (trail_segment of the
// oracle), then the geometry of make_mask.  One workgroup; keys (position t as an ordered integer, raster index) are
// sorted by a bitonic network in LDS.
__global__ __launch_bounds__(1024) void k_trail_segment(const uint32_t* __restrict__ list, const int32_t* __restrict__ cnt, uint32_t cap, int nyb,
                                                        int nxb, int nrho, sat_state* st, int32_t* err) {
    extern __shared__ unsigned long long seg_lds[];
    unsigned long long* K = seg_lds;                                   // [SAT_SEG_CAP]
    uint32_t* P = reinterpret_cast<uint32_t*>(K + SAT_SEG_CAP);        // [SAT_SEG_CAP]
    __shared__ int m_cnt, n2s;
    if (!st->accept) return;
    if (threadIdx.x == 0) m_cnt = 0;
    __syncthreads();
    const uint32_t n = min((uint32_t)*cnt, cap);
    const double c = st->c, s = st->s;
    const int off = nrho / 2, rcell = (int)st->rho + off;
    for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
        const uint32_t p = list[e];
        const double y = (double)(p / nxb), x = (double)(p % nxb);
        const double r = c * x + s * y;
        if ((int)(r > 0.0 ? r + 0.5 : r - 0.5) + off == rcell) {
            const int k = atomicAdd(&m_cnt, 1);
            if (k < SAT_SEG_CAP) { K[k] = dkey(y * c - x * s); P[k] = p; }
        }
    }
    __syncthreads();
    const int m = m_cnt;
    if (m > SAT_SEG_CAP || m < 2) {
        if (threadIdx.x == 0) { st->accept = 0; if (m > SAT_SEG_CAP) atomicOr(err, BBX_DERR_LIST_OVERFLOW); }
        return;
    }
    if (threadIdx.x == 0) { int n2 = 2; while (n2 < m) n2 <<= 1; n2s = n2; }
    __syncthreads();
    const int n2 = n2s;
    for (int i = m + threadIdx.x; i < n2; i += blockDim.x) { K[i] = ~0ull; P[i] = 0xffffffffu; }
    __syncthreads();
    for (int k = 2; k <= n2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n2; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long ka = K[i], kb = K[l]; const uint32_t pa = P[i], pb = P[l];
                    const bool a_gt_b = ka > kb || (ka == kb && pa > pb);
                    if (((i & k) == 0) == a_gt_b) { K[i] = kb; K[l] = ka; P[i] = pb; P[l] = pa; }
                }
            }
            __syncthreads();
        }
    if (threadIdx.x != 0) return;
    // runs
    int best0 = 0, best1 = 0, r0 = 0; double bestlen = -1.0;
    double tprev = dkey_inv(K[0]), tstart = tprev;
    for (int i = 1; i <= m; i++) {
        const double t = i < m ? dkey_inv(K[i]) : 0.0;
        if (i == m || t - tprev > 75.0) {
            const double len = tprev - tstart;
            if (len > bestlen) { bestlen = len; best0 = r0; best1 = i - 1; }
            r0 = i; tstart = t;
        }
        tprev = t;
    }
    const int px = (int)(P[best0] % nxb), py = (int)(P[best0] / nxb), qx = (int)(P[best1] % nxb), qy = (int)(P[best1] / nxb);
    const double theta_deg = 2.0 + 0.5 * (double)st->k;
    bool first_is_p;
    if (45.0 < theta_deg && theta_deg < 135.0) first_is_p = px < qx || (px == qx && py <= qy);
    else if (theta_deg <= 45.0) first_is_p = py > qy || (py == qy && px <= qx);
    else first_is_p = py < qy || (py == qy && px <= qx);
    const int x0 = first_is_p ? px : qx, y0 = first_is_p ? py : qy, x1 = first_is_p ? qx : px, y1 = first_is_p ? qy : py;
    st->seg[0] = x0; st->seg[1] = y0; st->seg[2] = x1; st->seg[3] = y1;
    if (x0 == x1 && y0 == y1) { st->accept = 0; return; }
    // geometry of make_mask
    const float top = fkey_inv(st->bmax_key), bmin = fkey_inv(st->bmin_key);
    if (!(top > 0.f)) { st->accept = 0; return; }                           // "Image has no positive values"
    float lof = bmin / top; if (lof < 0.f) lof = 0.f;
    st->top = (double)top; st->clip_lo = (double)lof; st->clip_hi = (double)(top / top);
    const double ddx = (double)(x1 - x0), ddy = (double)(y1 - y0);
    sat_rotate_matrix(nyb, nxb, ddx, ddy, st->m, &st->rot_rows, &st->rot_cols);
    const double ax = (double)x0 - st->m[2], ay = (double)y0 - st->m[5];
    st->sx = st->m[0] * ax + st->m[3] * ay; st->sy = st->m[1] * ax + st->m[4] * ay;      // to_output
    sat_rotate_matrix(st->rot_rows, st->rot_cols, ddx, -ddy, st->m2, &st->back_rows, &st->back_cols);
    st->crop_y = (int)((double)(st->back_rows - nyb) / 2.0); st->crop_x = (int)((double)(st->back_cols - nxb) / 2.0);
    int r0b = (int)st->sy - SAT_BANDH; if (r0b < 0) r0b = 0; if (r0b > st->rot_rows) r0b = st->rot_rows;
    int r1b = (int)st->sy + SAT_BANDH; if (r1b > st->rot_rows) r1b = st->rot_rows; if (r1b < r0b) r1b = r0b;
    st->band_r0 = r0b; st->band_rows = r1b - r0b;
    st->seg_ok = 1;
}

__device__ __forceinline__ double sat_cubic(double x, double f0, double f1, double f2, double f3) {
    return f1 + 0.5 * x * (f2 - f0 + x * (2.0 * f0 - 5.0 * f1 + 4.0 * f2 - f3 + x * (3.0 * (f1 - f2) + f3 - f0)));
}

// the band of the rotated image: skimage's bicubic interpolation (rows of the 4 x 4 patch first), 0 outside, then warp's clipping
__global__ __launch_bounds__(256) void k_sat_rotband(const float* __restrict__ b, int nyb, int nxb, const sat_state* __restrict__ st,
                                                     double* __restrict__ band, int band_cap_cols) {
    if (!st->seg_ok) return;
    const int cols = st->rot_cols, rows = st->band_rows;
    if (cols > band_cap_cols) return;
    const int c = blockIdx.x * blockDim.x + threadIdx.x, br = blockIdx.y;
    if (c >= cols || br >= rows) return;
    const double cf = (double)c, rf = (double)(st->band_r0 + br);
    const double ci = st->m[0] * cf + st->m[1] * rf + st->m[2];
    const double ri = st->m[3] * cf + st->m[4] * rf + st->m[5];
    const double fr0 = floor(ri), fc0 = floor(ci);
    const double xr = ri - fr0, xc = ci - fc0;
    const long long r0 = (long long)fr0 - 1, c0 = (long long)fc0 - 1;
    const float top = (float)st->top;
    double fr[4];
#pragma unroll
    for (int pr = 0; pr < 4; pr++) {
        double f[4];
#pragma unroll
        for (int pc = 0; pc < 4; pc++) {
            const long long rr = r0 + pr, cc = c0 + pc;
            double v = 0.0;
            if (rr >= 0 && rr < nyb && cc >= 0 && cc < nxb) { float q = b[(size_t)rr * nxb + cc] / top; if (q < 0.f) q = 0.f; v = (double)q; }
            f[pc] = v;
        }
        fr[pr] = sat_cubic(xc, f[0], f[1], f[2], f[3]);
    }
    double out = sat_cubic(xr, fr[0], fr[1], fr[2], fr[3]);
    const double lo = st->clip_lo, hi = st->clip_hi;
    if (!(lo <= 0.0 && 0.0 <= hi) && out == 0.0) out = 0.0;
    else out = fmin(fmax(out, lo), hi);
    band[(size_t)br * band_cap_cols + c] = out;
}

// ---- statistics of a window's row medians (thread 0): the oracle's sigma_clipped_mean / biweight_midvariance -------
__device__ void sat_sort(double* a, int n) { for (int i = 1; i < n; i++) { const double v = a[i]; int j = i - 1; while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; j--; } a[j + 1] = v; } }
__device__ double sat_median(const double* a, int n) {
    double t[2 * SAT_SUBLEN + 2];
    for (int i = 0; i < n; i++) t[i] = a[i];
    sat_sort(t, n);
    return (n & 1) ? t[n / 2] : (t[n / 2 - 1] + t[n / 2]) / 2.0;
}
__device__ double sat_np_sum(const double* a, int n) {
    if (n < 8) { double r = 0.0; for (int i = 0; i < n; i++) r = r + a[i]; return r; }
    double r[8];
    for (int k = 0; k < 8; k++) r[k] = a[k];
    int i = 8;
    for (; i < n - (n % 8); i += 8) for (int k = 0; k < 8; k++) r[k] = r[k] + a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res = res + a[i];
    return res;
}
__device__ double sat_clipped_mean(const double* x, int n0) {
    double v[2 * SAT_SUBLEN + 2]; int n = 0;
    for (int i = 0; i < n0; i++) if (isfinite(x[i])) v[n++] = x[i];
    for (int it = 0; it < 5; it++) {
        if (n == 0) break;
        const double cen = sat_median(v, n);
        double s = 0.0; for (int i = 0; i < n; i++) s += v[i];
        const double mu = s / n;
        double q = 0.0; for (int i = 0; i < n; i++) q += (v[i] - mu) * (v[i] - mu);
        const double sd = sqrt(q / n);
        const double lo = cen - sd * 3.0, hi = cen + sd * 3.0;
        int k = 0; for (int i = 0; i < n; i++) if (v[i] >= lo && v[i] <= hi) v[k++] = v[i];
        if (k == n) break;
        n = k;
    }
    if (n == 0) return __builtin_nan("");
    double s = 0.0; for (int i = 0; i < n; i++) s += v[i];
    return s / n;
}
__device__ double sat_midvariance(const double* x, int n) {
    double d[2 * SAT_SUBLEN + 2], a[2 * SAT_SUBLEN + 2], f1[2 * SAT_SUBLEN + 2], f2[2 * SAT_SUBLEN + 2];
    const double med = sat_median(x, n);
    for (int i = 0; i < n; i++) { d[i] = x[i] - med; a[i] = fabs(x[i] - med); }
    const double mad = sat_median(a, n);
    if (mad == 0.0) return 0.0;
    for (int i = 0; i < n; i++) {
        double u = d[i] / (9.0 * mad);
        const bool inside = fabs(u) < 1.0;
        u = u * u;
        const double w = 1.0 - u, w2 = w * w;
        f1[i] = inside ? d[i] * d[i] * (w2 * w2) : 0.0;
        f2[i] = inside ? (1.0 - u) * (1.0 - 5.0 * u) : 0.0;
    }
    const double s2 = fabs(sat_np_sum(f2, n));
    return (double)n * sat_np_sum(f1, n) / (s2 * s2);
}

// The walk along the trail (make_mask's loop): one workgroup; a window's rows sit in LDS, their medians by rank counting.
__global__ __launch_bounds__(256) void k_sat_walk(const double* __restrict__ band, int band_cap_cols, sat_state* st, sat_rect* __restrict__ rects,
                                                  int32_t* err) {
    if (!st->seg_ok) return;
    __shared__ double L[2 * SAT_SUBLEN][SAT_SUBW];
    __shared__ double med_lo[2 * SAT_SUBLEN], med_hi[2 * SAT_SUBLEN];
    __shared__ int box[4], go, zmin, zmax;
    const int tid = threadIdx.x, dxw = SAT_SUBW / 2;
    const int rot_rows = st->rot_rows, rot_cols = st->rot_cols;
    if (rot_cols > band_cap_cols) { if (tid == 0) { st->seg_ok = 0; atomicOr(err, BBX_DERR_NOTCONV); } return; }
    const double sx = st->sx, sy = st->sy;
    int centre0 = 0, nwin = 0, rmin = 0x7fffffff, rmax = -1;
    bool failed = false;
    // phase 0: the first look at (sx, sy); phase 1: to the right; phase 2: to the left
    for (int phase = 0; phase < 3 && !failed; phase++) {
        double nextx = phase == 0 ? sx : (phase == 1 ? sx + (double)dxw : sx - (double)dxw);
        double cy = phase == 0 ? sy : (double)centre0;
        for (int it = 0; it < (phase == 0 ? 1 : 500); it++) {
            if (tid == 0) {
                double fx0 = nextx - dxw, fx1 = nextx + dxw, fy0 = cy - SAT_SUBLEN, fy1 = cy + SAT_SUBLEN;
                fx0 = fx0 > 0.0 ? fx0 : 0.0; fy0 = fy0 > 0.0 ? fy0 : 0.0;
                fx1 = fx1 < (double)rot_cols ? fx1 : (double)rot_cols; fy1 = fy1 < (double)rot_rows ? fy1 : (double)rot_rows;
                go = (fy1 <= fy0 || fx1 <= fx0) ? 0 : 1;                       // IndexError: the window left the frame
                box[0] = (int)fx0; box[1] = (int)fx1; box[2] = (int)fy0; box[3] = (int)fy1;
                if (go && (box[2] < st->band_r0 || box[3] > st->band_r0 + st->band_rows)) { go = -1; }      // left the interpolated band
                if (go && (box[1] <= box[0] || box[3] <= box[2])) go = 0;
            }
            __syncthreads();
            if (go <= 0) { if (go < 0 || phase == 0) failed = true; break; }
            const int ix0 = box[0], ix1 = box[1], iy0 = box[2], iy1 = box[3], nr = iy1 - iy0, nc = ix1 - ix0;
            for (int e = tid; e < nr * nc; e += blockDim.x) {
                const int r = e / nc, c = e - r * nc;
                L[r][c] = band[(size_t)(iy0 + r - st->band_r0) * band_cap_cols + ix0 + c];
            }
            __syncthreads();
            for (int r = 0; r < nr; r++) {
                if (tid < nc) {
                    const double v = L[r][tid];
                    int rank = 0;
                    for (int j = 0; j < nc; j++) { const double w = L[r][j]; rank += (w < v || (w == v && j < tid)) ? 1 : 0; }
                    if (rank == (nc - 1) / 2) med_lo[r] = v;
                    if (rank == nc / 2) med_hi[r] = v;
                }
            }
            __syncthreads();
            if (tid == 0) {
                double medarr[2 * SAT_SUBLEN];
                for (int r = 0; r < nr; r++) medarr[r] = (nc & 1) ? med_lo[r] : (med_lo[r] + med_hi[r]) / 2.0;
                const double mean = sat_clipped_mean(medarr, nr), var = sat_midvariance(medarr, nr);
                const double thr = mean + 5.0 * var;
                int lo = -1, hi = -1;
                for (int r = 0; r < nr; r++) if (medarr[r] > thr) { if (lo < 0) lo = r; hi = r; }
                zmin = lo; zmax = hi;
            }
            __syncthreads();
            if (phase == 0 && (nr <= SAT_SUBLEN || zmin < 0)) { failed = true; break; }      // make_mask's ValueErrors: no mask
            if (zmin < 0) break;                                                   // no trail in this window
            // paint (pad = 0)
            int r0 = iy0 + zmin; if (r0 < 0) r0 = 0;
            int r1 = iy0 + zmax + 1; if (r1 > rot_rows) r1 = rot_rows;
            if (nwin >= SAT_MAXWIN) { failed = true; break; }
            if (tid == 0) { rects[nwin].r0 = r0; rects[nwin].r1 = r1; rects[nwin].c0 = ix0; rects[nwin].c1 = ix1; }
            nwin++;
            rmin = r0 < rmin ? r0 : rmin; rmax = r1 > rmax ? r1 : rmax;
            const int centre = iy0 + (int)ceil((double)zmin + (double)(zmax - zmin) / 2.0);
            if (phase == 0) centre0 = centre;
            cy = (double)centre;
            nextx += phase == 2 ? -(double)dxw : (double)dxw;
            __syncthreads();
        }
        __syncthreads();
    }
    if (tid == 0) {
        if (failed && nwin == 0) { st->found = 0; st->nwin = 0; }
        else if (failed) { st->found = 0; st->nwin = 0; atomicOr(err, BBX_DERR_NOTCONV); }
        else { st->found = nwin > 0 ? 1 : 0; st->nwin = nwin; st->row_min = rmin; st->row_max = rmax; }
    }
}

// the strips rotated back (skimage.transform.rotate, order 1, cropped about the centre) and made boolean; 2 x 2 un-binning
__global__ __launch_bounds__(256) void k_sat_paint(uint8_t* mask, int nyb, int nxb, const sat_state* __restrict__ st, const sat_rect* __restrict__ rects) {
    if (!st->found) return;
    __shared__ sat_rect R[SAT_MAXWIN];
    const int nwin = st->nwin;
    for (int i = threadIdx.x; i < nwin; i += blockDim.x) R[i] = rects[i];
    __syncthreads();
    const int rot_rows = st->rot_rows, rot_cols = st->rot_cols, rmin = st->row_min, rmax = st->row_max;
    const double m0 = st->m2[0], m1 = st->m2[1], m2 = st->m2[2], m3 = st->m2[3], m4 = st->m2[4], m5 = st->m2[5];
    const int cy = st->crop_y, cx = st->crop_x;
    const size_t n = (size_t)nyb * nxb, nx = (size_t)nxb * 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int Y = (int)(i / nxb), X = (int)(i - (size_t)Y * nxb);
        const double rf = (double)(cy + Y), cf = (double)(cx + X);
        const double c = m0 * cf + m1 * rf + m2;
        const double r = m3 * cf + m4 * rf + m5;
        if (r < (double)rmin - 2.0 || r > (double)rmax + 2.0) continue;
        const double fminr = floor(r), fminc = floor(c);
        const long long minr = (long long)fminr, minc = (long long)fminc, maxr = (long long)ceil(r), maxc = (long long)ceil(c);
        const double dr = r - fminr, dc = c - fminc;
        double p[4];
        const long long rr[4] = {minr, minr, maxr, maxr}, cc[4] = {minc, maxc, minc, maxc};
        for (int q = 0; q < 4; q++) {
            double v = 0.0;
            if (rr[q] >= 0 && rr[q] < rot_rows && cc[q] >= 0 && cc[q] < rot_cols)
                for (int w = 0; w < nwin; w++)
                    if (rr[q] >= R[w].r0 && rr[q] < R[w].r1 && cc[q] >= R[w].c0 && cc[q] < R[w].c1) { v = 1.0; break; }
            p[q] = v;
        }
        const double top = (1 - dc) * p[0] + dc * p[1], bot = (1 - dc) * p[2] + dc * p[3];
        const double out = (1 - dr) * top + dr * bot;
        if (out != 0.0) {
            uint16_t* a = reinterpret_cast<uint16_t*>(mask + (size_t)(2 * Y) * nx + 2 * X);
            uint16_t* bq = reinterpret_cast<uint16_t*>(mask + (size_t)(2 * Y + 1) * nx + 2 * X);
            *a |= (uint16_t)(BBX_MASK_SATELLITE | (BBX_MASK_SATELLITE << 8));
            *bq |= (uint16_t)(BBX_MASK_SATELLITE | (BBX_MASK_SATELLITE << 8));
        }
    }
}

__global__ void k_sat_info(const sat_state* __restrict__ st, float* info) {
    if (threadIdx.x == 0) {
        info[0] = fkey_inv(st->bmax_key); info[1] = fkey_inv(st->bmin_key); info[2] = (float)st->votes; info[3] = (float)st->k;
        info[4] = (float)st->rho; info[5] = (float)st->nwin; info[6] = (float)st->seg_ok; info[7] = (float)st->found;
    }
}

extern "C" int bbx_count_objects(bbx_ctx* ctx, int ny, int nx, const uint8_t* d_mask, int bit, int32_t* d_count, void* stream);

int bbx_canny_edges(bbx_ctx* ctx, const float* d_bin, int ny, int nx, const double* h_gauss, int radius, double low_frac, double high_frac,
                    int min_size, uint32_t* d_out, int32_t* d_out_cnt, uint32_t out_cap, hipStream_t s);

extern "C" int bbx_sat_trails(bbx_ctx* ctx, int ny, int nx, const float* d_data, uint8_t* d_mask, const double* h_cos_sin,
                              int ntheta, const double* h_gauss, int gauss_radius, int32_t* d_nsats, float* d_info, void* stream) {
    if (!ctx || !d_data || !d_mask || !h_cos_sin || !h_gauss || !d_nsats || !d_info || ny < 8 || nx < 8 || (ny & 1) || (nx & 1) ||
        ntheta < 4 || ntheta > 4096 || ((uintptr_t)d_data) % 8)
        return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nyb = ny / 2, nxb = nx / 2;
    const size_t nb = (size_t)nyb * nxb;
    // skimage.transform.hough_line: offset = ceil(hypot(ny, nx)), 2 offset cells of one pixel
    const int nrho = 2 * (int)ceil(sqrt((double)nyb * nyb + (double)nxb * nxb));
    int rc;
    const size_t cap = nb / 8 + 4096;
    // workspace: binned frame | edge list | cos/sin | partials | state
    const size_t o_bin = 0, o_list = o_bin + ((nb * 4 + 255) & ~(size_t)255);
    const size_t o_cs = o_list + ((cap * 4 + 255) & ~(size_t)255), o_part = o_cs + (((size_t)ntheta * 16 + 255) & ~(size_t)255);
    const size_t o_st = o_part + SAT_BLOCKS * 3 * 8;
    // make_mask: the strips of the walk and the band of the rotated frame (its width is at most the frame's diagonal + 2)
    const int band_cols = nrho / 2 + 8;
    const size_t o_rect = (o_st + sizeof(sat_state) + 255) & ~(size_t)255, o_band = (o_rect + SAT_MAXWIN * sizeof(sat_rect) + 255) & ~(size_t)255;
    const size_t total = o_band + (size_t)2 * SAT_BANDH * band_cols * sizeof(double) + 256;
    char* ws = (char*)bbx_ws(ctx, WS_CAND, total, &rc); if (rc) return rc;
    float* bin = (float*)(ws + o_bin); uint32_t* list = (uint32_t*)(ws + o_list);
    double* cs = (double*)(ws + o_cs); sat_state* st = (sat_state*)(ws + o_st);
    sat_rect* rects = (sat_rect*)(ws + o_rect); double* band = (double*)(ws + o_band);
    int32_t* cnt = &ctx->d_counters[CNT_CAND];
    BBX_HIP(hipMemcpyAsync(cs, h_cos_sin, (size_t)ntheta * 16, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_sat_init, dim3(1), dim3(128), 0, s, st);
    hipLaunchKernelGGL(k_bin2, dim3((nxb + 255) / 256, nyb), dim3(256), 0, s, d_data, nyb, nxb, bin);
    hipLaunchKernelGGL(k_bin_minmax, dim3(1024), dim3(256), 0, s, bin, nb, st);
    // acstools' front end: sigma = 3 (sat_detect), low_thresh = 0.1 (default), h_thresh = 0.2 (sat_detect), small_edge = 60
    rc = bbx_canny_edges(ctx, bin, nyb, nxb, h_gauss, gauss_radius, 0.1, 0.2, 60, list, cnt, (uint32_t)cap, s); if (rc) return rc;
#ifdef SATV                                                        // timing knock-outs (tools/exp/satvar.sh; never in the product build)
    if (getenv("BBX_DBG_SAT") && atoi(getenv("BBX_DBG_SAT")) >= 1) return BBX_OK;
#endif
    if ((size_t)nrho * 4 <= 150 * 1024) {
        // rho histogram of one angle fits in LDS (frames up to ~26k binned pixels across)
        hipLaunchKernelGGL(k_hough_lds, dim3(ntheta), dim3(1024), (size_t)nrho * 4, s, list, cnt, (uint32_t)cap, nxb, cs, nrho, st);
    } else {
        return BBX_ERR_ARG;                                  // frame too large for the LDS histogram
    }
    hipLaunchKernelGGL(k_trail_decide, dim3(1), dim3(64), 0, s, st, cs, nrho, nyb, nxb, 210);    // the probabilistic transform's threshold
    const double refine_tol[4] = {0.0, 3.0, 2.0, 1.5};                                          // oracle/sattrail.py REFINE_TOL
    for (int it = 0; it < 4; it++) {
        hipLaunchKernelGGL(k_trail_cone, dim3(256), dim3(256), 0, s, list, cnt, (uint32_t)cap, nxb, st, refine_tol[it]);
        hipLaunchKernelGGL(k_trail_refine, dim3(1), dim3(64), 0, s, st);
    }
    hipLaunchKernelGGL(k_trail_support, dim3(256), dim3(256), 0, s, list, cnt, (uint32_t)cap, nxb, st);
    hipLaunchKernelGGL(k_trail_accept, dim3(1), dim3(64), 0, s, st, 40.0);                       // buf = 40 (sat_detect)
    // make_mask (oracle/sattrail.py): segment + geometry, the band of the rotated frame, the walk, the strips painted back
    const size_t seg_lds = (size_t)SAT_SEG_CAP * 12;
    if (ctx->sat_attr_set != 1) {
        BBX_HIP(hipFuncSetAttribute((const void*)k_trail_segment, hipFuncAttributeMaxDynamicSharedMemorySize, (int)seg_lds));
        ctx->sat_attr_set = 1;
    }
    hipLaunchKernelGGL(k_trail_segment, dim3(1), dim3(1024), seg_lds, s, list, cnt, (uint32_t)cap, nyb, nxb, nrho, st, ctx->d_err);
    hipLaunchKernelGGL(k_sat_rotband, dim3((band_cols + 255) / 256, 2 * SAT_BANDH), dim3(256), 0, s, bin, nyb, nxb, st, band, band_cols);
    hipLaunchKernelGGL(k_sat_walk, dim3(1), dim3(256), 0, s, band, band_cols, st, rects, ctx->d_err);
    hipLaunchKernelGGL(k_sat_paint, dim3(2048), dim3(256), 0, s, d_mask, nyb, nxb, st, rects);
    hipLaunchKernelGGL(k_sat_info, dim3(1), dim3(64), 0, s, st, d_info);
    BBX_LAUNCH_CHECK();
    return bbx_count_objects(ctx, ny, nx, d_mask, BBX_MASK_SATELLITE, d_nsats, stream);
}

// bbx_build_flags (bbx_ctx.hip): the satellite stage's timing knock-outs (this file, bbx_canny.hip) compiled in?
int bbx_build_flags_sat(void) {
#ifdef SATV
    return 8;
#else
    return 0;
#endif
}
