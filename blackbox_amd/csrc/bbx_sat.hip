// bbx_sat.hip -- satellite-trail masking (reference sat_detect, blackbox.py:4163-4254).
// The reference calls acstools.satdet (or ASTA, a CNN without weights here); acstools is absent from this
// image.  The detector follows acstools' published chain where it is deterministic and is specified in
// oracle/sattrail.py:
//
//   2x2 sum binning -> [bbx_canny.hip: percentile rescale, Canny(sigma 3, 0.1 / 0.2 of the maximum),
//   remove_small_objects(60): pinned against scikit-image] -> full Hough accumulator over acstools' theta grid
//   (2 .. 177.5 deg, skimage.transform.hough_line's cells; stands in for the *random* probabilistic transform)
//   -> best cell with >= 210 votes whose supporting edge pixels span the frame to within buf = 40 px of both
//   borders -> perpendicular profile (clipped level / sigma of the binned frame, float64 atomics into 81 bins)
//   -> strip -> bit 16.
// Traffic: one read of the frame (4N) for binning, then everything works on the 4x smaller binned frame and on
// sparse pixel lists.
#include "bbx_common.h"

#define SAT_PROF_HALF 40
#define SAT_NPROF (2 * SAT_PROF_HALF + 1)
#define SAT_BLOCKS 1024

struct sat_state {
    double lo, hi, mean, std;             // clipped statistics of the binned frame
    unsigned long long best;              // (votes << 32) | (0xffffffff - flat index)
    int accept, found, k, lo_off, hi_off; // line accepted / strip found, theta index, strip offsets
    int votes;
    double rho, c, s, chord;
    unsigned long long tmin, tmax;        // ordered keys of the smallest / largest position of a supporting edge pixel along the line
    double t0, t1;                        // where the line enters / leaves the frame
    unsigned long long fit_n; long long fit_t, fit_d, fit_tt, fit_td;   // integer sums (1/16 px) of the refinement
    double icpt, slope;                   // refined line: distance d - (icpt + slope t)
    double prof_sum[SAT_NPROF];
    unsigned long long prof_n[SAT_NPROF];
};

__global__ __launch_bounds__(256) void k_bin2(const float* __restrict__ d, int nyb, int nxb, float* __restrict__ b) {
    const int X = blockIdx.x * blockDim.x + threadIdx.x, Y = blockIdx.y;
    if (X >= nxb) return;
    const size_t nx = (size_t)nxb * 2;
    const float2 r0 = *(const float2*)(d + (size_t)(2 * Y) * nx + 2 * X);
    const float2 r1 = *(const float2*)(d + (size_t)(2 * Y + 1) * nx + 2 * X);
    b[(size_t)Y * nxb + X] = (r0.x + r0.y) + (r1.x + r1.y);
}

__global__ void k_sat_init(sat_state* st) {
    if (threadIdx.x == 0) {
        st->lo = -__builtin_huge_val(); st->hi = __builtin_huge_val(); st->mean = 0; st->std = 0; st->best = 0;
        st->accept = 0; st->found = 0; st->k = 0; st->lo_off = 0; st->hi_off = 0; st->votes = 0;
        st->rho = 0; st->c = 0; st->s = 0; st->chord = 0; st->tmin = ~0ull; st->tmax = 0ull; st->t0 = 0; st->t1 = 0;
        st->fit_n = 0; st->fit_t = st->fit_d = st->fit_tt = st->fit_td = 0; st->icpt = 0; st->slope = 0;
    }
    if (threadIdx.x < SAT_NPROF) { st->prof_sum[threadIdx.x] = 0.0; st->prof_n[threadIdx.x] = 0; }
}

__device__ __forceinline__ void clip_acc(float f, double lo, double hi, double& s1, double& s2, int& cnt) {
    const double x = (double)f;
    if (isfinite(f) && x >= lo && x <= hi) { s1 += x; s2 += x * x; cnt++; }
}

// sums of the binned pixels inside the current clip range: 16-byte loads, four in flight per
// thread; one partial triple per workgroup (fixed launch shape -> the same sums in every run)
__global__ __launch_bounds__(256) void k_clip_pass(const float* __restrict__ b, size_t n, const sat_state* __restrict__ st,
                                                   double* __restrict__ partial) {
    const double lo = st->lo, hi = st->hi;
    double s1 = 0.0, s2 = 0.0; int cnt = 0;
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * blockDim.x;
    const float4* b4 = (const float4*)b;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; u++) q[u] = b4[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            clip_acc(q[u].x, lo, hi, s1, s2, cnt); clip_acc(q[u].y, lo, hi, s1, s2, cnt);
            clip_acc(q[u].z, lo, hi, s1, s2, cnt); clip_acc(q[u].w, lo, hi, s1, s2, cnt);
        }
    }
    for (; i < n4; i += stride) {
        const float4 q = b4[i];
        clip_acc(q.x, lo, hi, s1, s2, cnt); clip_acc(q.y, lo, hi, s1, s2, cnt);
        clip_acc(q.z, lo, hi, s1, s2, cnt); clip_acc(q.w, lo, hi, s1, s2, cnt);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) clip_acc(b[(n4 << 2) + threadIdx.x], lo, hi, s1, s2, cnt);
    __shared__ double sh1[4], sh2[4]; __shared__ int shn[4];
    s1 = wave_sum_f64(s1); s2 = wave_sum_f64(s2); cnt = wave_sum_i32(cnt);
    if ((threadIdx.x & 63) == 0) { sh1[threadIdx.x >> 6] = s1; sh2[threadIdx.x >> 6] = s2; shn[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* p = partial + (size_t)blockIdx.x * 3;
        p[0] = (sh1[0] + sh1[1]) + (sh1[2] + sh1[3]); p[1] = (sh2[0] + sh2[1]) + (sh2[2] + sh2[3]);
        p[2] = (double)(shn[0] + shn[1] + shn[2] + shn[3]);
    }
}

// one workgroup of 256 folds the partial triples (thread t takes blocks t, t+256, ...; then a
// fixed tree) and narrows the clip range
__global__ __launch_bounds__(256) void k_clip_update(sat_state* st, const double* __restrict__ partial, int nblocks) {
    __shared__ double sh[3][4];
    double a = 0, q = 0, m = 0;
    for (int b = threadIdx.x; b < nblocks; b += 256) { a += partial[3 * b]; q += partial[3 * b + 1]; m += partial[3 * b + 2]; }
    a = wave_sum_f64(a); q = wave_sum_f64(q); m = wave_sum_f64(m);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = q; sh[2][threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    a = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]); q = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    m = (sh[2][0] + sh[2][1]) + (sh[2][2] + sh[2][3]);
    const double mean = a / m;
    double var = q / m - mean * mean; if (var < 0) var = 0;
    const double sd = sqrt(var);
    st->mean = mean; st->std = sd;
    const double lo = mean - 3.0 * sd, hi = mean + 3.0 * sd;
    if (lo > st->lo) st->lo = lo;
    if (hi < st->hi) st->hi = hi;
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

__global__ __launch_bounds__(256) void k_trail_profile(const float* __restrict__ b, int nyb, int nxb, sat_state* st) {
    if (!st->accept) return;
    __shared__ double lsum[SAT_NPROF];
    __shared__ unsigned long long lcnt[SAT_NPROF];
    for (int i = threadIdx.x; i < SAT_NPROF; i += blockDim.x) { lsum[i] = 0.0; lcnt[i] = 0; }
    __syncthreads();
    const double c = st->c, s = st->s, rho = st->rho, t1 = st->mean + 50.0 * st->std, icpt = st->icpt, slope = st->slope;
    const size_t n = (size_t)nyb * nxb;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / nxb), x = (int)(i - (size_t)y * nxb);
        const double dd = (double)x * c + (double)y * s - rho, tt = (double)y * c - (double)x * s;
        const long long d = (long long)floor(dd - (icpt + slope * tt) + 0.5);
        if (d < -SAT_PROF_HALF || d > SAT_PROF_HALF) continue;
        const float f = b[i];
        if (!isfinite(f) || !((double)f < t1)) continue;
        atomicAdd(&lsum[d + SAT_PROF_HALF], (double)f);
        atomicAdd(&lcnt[d + SAT_PROF_HALF], 1ull);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SAT_NPROF; i += blockDim.x) {
        if (lcnt[i]) { atomicAdd(&st->prof_sum[i], lsum[i]); atomicAdd(&st->prof_n[i], lcnt[i]); }
    }
}

__global__ void k_trail_strip(sat_state* st) {
    if (threadIdx.x != 0 || !st->accept) return;
    double prof[SAT_NPROF]; double pmax = -__builtin_huge_val(); int ipk = 0;
    for (int i = 0; i < SAT_NPROF; i++) {
        prof[i] = (st->prof_n[i] > 0 ? st->prof_sum[i] / (double)st->prof_n[i] : st->mean) - st->mean;
        if (prof[i] > pmax) { pmax = prof[i]; ipk = i; }
    }
    bool above[SAT_NPROF];
    for (int i = 0; i < SAT_NPROF; i++) {
        const double nn = st->prof_n[i] > 0 ? (double)st->prof_n[i] : 1.0;
        const double thr = fmax(5.0 * st->std / sqrt(nn), 0.1 * pmax);
        above[i] = (prof[i] > thr) && st->prof_n[i] > 0;
    }
    if (!above[ipk]) return;
    int lo = ipk, hi = ipk;
    while (lo - 1 >= 0 && above[lo - 1]) lo--;
    while (hi + 1 < SAT_NPROF && above[hi + 1]) hi++;
    st->lo_off = lo - SAT_PROF_HALF; st->hi_off = hi - SAT_PROF_HALF; st->found = 1;
}

__global__ __launch_bounds__(256) void k_trail_mask(uint8_t* mask, int ny, int nx, const sat_state* __restrict__ st) {
    if (!st->found) return;
    const double c = st->c, s = st->s, rho = st->rho, icpt = st->icpt, slope = st->slope;
    const int lo = st->lo_off, hi = st->hi_off;
    const size_t n = (size_t)ny * nx;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int Y = (int)(i / nx), X = (int)(i - (size_t)Y * nx);
        const double xb = (double)(X >> 1), yb = (double)(Y >> 1);
        const double dd = xb * c + yb * s - rho, tt = yb * c - xb * s;
        const long long d = (long long)floor(dd - (icpt + slope * tt) + 0.5);
        if (d >= lo && d <= hi) mask[i] |= BBX_MASK_SATELLITE;
    }
}

__global__ void k_sat_info(const sat_state* __restrict__ st, float* info) {
    if (threadIdx.x == 0) {
        info[0] = (float)st->mean; info[1] = (float)st->std; info[2] = (float)st->votes; info[3] = (float)st->k;
        info[4] = (float)st->rho; info[5] = (float)st->lo_off; info[6] = (float)st->hi_off; info[7] = (float)st->found;
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
    const size_t o_st = o_part + SAT_BLOCKS * 3 * 8, total = o_st + sizeof(sat_state) + 256;
    char* ws = (char*)bbx_ws(ctx, WS_CAND, total, &rc); if (rc) return rc;
    float* bin = (float*)(ws + o_bin); uint32_t* list = (uint32_t*)(ws + o_list);
    double* cs = (double*)(ws + o_cs); double* partial = (double*)(ws + o_part); sat_state* st = (sat_state*)(ws + o_st);
    int32_t* cnt = &ctx->d_counters[CNT_CAND];
    BBX_HIP(hipMemcpyAsync(cs, h_cos_sin, (size_t)ntheta * 16, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_sat_init, dim3(1), dim3(128), 0, s, st);
    hipLaunchKernelGGL(k_bin2, dim3((nxb + 255) / 256, nyb), dim3(256), 0, s, d_data, nyb, nxb, bin);
    for (int pass = 0; pass < 4; pass++) {
        hipLaunchKernelGGL(k_clip_pass, dim3(SAT_BLOCKS), dim3(256), 0, s, bin, nb, st, partial);
        hipLaunchKernelGGL(k_clip_update, dim3(1), dim3(256), 0, s, st, partial, SAT_BLOCKS);
    }
    // acstools' front end: sigma = 3 (sat_detect), low_thresh = 0.1 (default), h_thresh = 0.2 (sat_detect), small_edge = 60
    rc = bbx_canny_edges(ctx, bin, nyb, nxb, h_gauss, gauss_radius, 0.1, 0.2, 60, list, cnt, (uint32_t)cap, s); if (rc) return rc;
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
    hipLaunchKernelGGL(k_trail_profile, dim3(2048), dim3(256), 0, s, bin, nyb, nxb, st);
    hipLaunchKernelGGL(k_trail_strip, dim3(1), dim3(64), 0, s, st);
    hipLaunchKernelGGL(k_trail_mask, dim3(2048), dim3(256), 0, s, d_mask, ny, nx, st);
    hipLaunchKernelGGL(k_sat_info, dim3(1), dim3(64), 0, s, st, d_info);
    BBX_LAUNCH_CHECK();
    return bbx_count_objects(ctx, ny, nx, d_mask, BBX_MASK_SATELLITE, d_nsats, stream);
}
