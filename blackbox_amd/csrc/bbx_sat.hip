// bbx_sat.hip -- satellite-trail masking (reference sat_detect, blackbox.py:4163-4254).
// [EXT / unpinnable: acstools' probabilistic Hough is random, ASTA is a CNN without weights;
//  the deterministic classical detector implemented here is specified in oracle/sattrail.py]
//
//   2x2 sum binning -> clipped level/sigma -> edge pixels (3..50 sigma) -> full Hough
//   accumulator (720 angles x 1-px rho bins, one integer atomic per vote) -> best cell ->
//   chord test -> perpendicular profile (float64 atomics into 81 bins) -> strip -> bit 16.
// Traffic: one read of the frame (4N) for binning, then everything works on the 4x smaller
// binned frame and on a sparse edge list.
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
        st->rho = 0; st->c = 0; st->s = 0; st->chord = 0;
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

// edge pixels -> compact list.  Hits are staged in LDS and flushed with one global atomic
// per block and flush (a single global counter hammered per wave serialises badly).
#define EDGE_STAGE 2048
__global__ __launch_bounds__(256) void k_edge_compact(const float* __restrict__ b, size_t n, const sat_state* __restrict__ st,
                                                      uint32_t* list, int32_t* cnt, uint32_t cap, int32_t* err) {
    __shared__ uint32_t stage[EDGE_STAGE];
    __shared__ unsigned scnt, sbase;
    const double t0 = st->mean + 3.0 * st->std, t1 = st->mean + 50.0 * st->std;
    if (threadIdx.x == 0) scnt = 0;
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t nend = ((n + stride - 1) / stride) * stride;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nend; i += stride) {
        bool hit = false;
        if (i < n) { const double x = (double)b[i]; hit = (x > t0 && x < t1); }
        if (hit) { const unsigned k = atomicAdd(&scnt, 1u); stage[k] = (uint32_t)i; }      // <= 256 per round, room checked below
        __syncthreads();
        const unsigned c = scnt;
        const bool last = (i + stride >= nend);
        if (c + 256 > EDGE_STAGE || last) {
            if (threadIdx.x == 0) sbase = c ? atomicAdd((unsigned*)cnt, c) : 0u;
            __syncthreads();
            for (unsigned k = threadIdx.x; k < c; k += blockDim.x) {
                const unsigned pos = sbase + k;
                if (pos < cap) list[pos] = stage[k]; else atomicOr(err, BBX_DERR_LIST_OVERFLOW);
            }
            __syncthreads();
            if (threadIdx.x == 0) scnt = 0;
        }
        __syncthreads();
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
        atomicAdd(&hist[(int)floor(x * c + y * s + 0.5) + off], 1u);
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

__global__ void k_trail_decide(sat_state* st, const double* __restrict__ cs, int nrho, int nyb, int nxb) {
    if (threadIdx.x != 0) return;
    const unsigned long long b = st->best;
    if (!b) return;
    const unsigned votes = (unsigned)(b >> 32);
    const unsigned long long flat = 0xffffffffull - (b & 0xffffffffull);
    const int k = (int)(flat / nrho), r = (int)(flat % nrho);
    st->k = k; st->votes = (int)votes; st->rho = (double)(r - nrho / 2); st->c = cs[2 * k]; st->s = cs[2 * k + 1];
    st->chord = chord_length(st->c, st->s, st->rho, nyb, nxb);
    st->accept = (votes >= 200u && (double)votes >= 0.2 * st->chord) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_trail_profile(const float* __restrict__ b, int nyb, int nxb, sat_state* st) {
    if (!st->accept) return;
    __shared__ double lsum[SAT_NPROF];
    __shared__ unsigned long long lcnt[SAT_NPROF];
    for (int i = threadIdx.x; i < SAT_NPROF; i += blockDim.x) { lsum[i] = 0.0; lcnt[i] = 0; }
    __syncthreads();
    const double c = st->c, s = st->s, rho = st->rho, t1 = st->mean + 50.0 * st->std;
    const size_t n = (size_t)nyb * nxb;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / nxb), x = (int)(i - (size_t)y * nxb);
        const long long d = (long long)floor((double)x * c + (double)y * s - rho + 0.5);
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
    const double c = st->c, s = st->s, rho = st->rho;
    const int lo = st->lo_off, hi = st->hi_off;
    const size_t n = (size_t)ny * nx;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int Y = (int)(i / nx), X = (int)(i - (size_t)Y * nx);
        const long long d = (long long)floor((double)(X >> 1) * c + (double)(Y >> 1) * s - rho + 0.5);
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

extern "C" int bbx_sat_trails(bbx_ctx* ctx, int ny, int nx, const float* d_data, uint8_t* d_mask, const double* h_cos_sin,
                              int ntheta, int32_t* d_nsats, float* d_info, void* stream) {
    if (!ctx || !d_data || !d_mask || !h_cos_sin || !d_nsats || !d_info || ny < 4 || nx < 4 || (ny & 1) || (nx & 1) ||
        ntheta < 4 || ntheta > 4096 || ((uintptr_t)d_data) % 8)
        return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int nyb = ny / 2, nxb = nx / 2;
    const size_t nb = (size_t)nyb * nxb;
    const int nrho = 2 * (int)ceil(hypot((double)nyb, (double)nxb)) + 1;
    int rc;
    const size_t cap = nb / 8 + 4096;
    // workspace: binned frame | accumulator | edge list | cos/sin | partials | state
    const size_t o_bin = 0, o_acc = o_bin + ((nb * 4 + 255) & ~(size_t)255), o_list = o_acc + (((size_t)ntheta * nrho * 4 + 255) & ~(size_t)255);
    const size_t o_cs = o_list + ((cap * 4 + 255) & ~(size_t)255), o_part = o_cs + (((size_t)ntheta * 16 + 255) & ~(size_t)255);
    const size_t o_st = o_part + SAT_BLOCKS * 3 * 8, total = o_st + sizeof(sat_state) + 256;
    char* ws = (char*)bbx_ws(ctx, WS_CAND, total, &rc); if (rc) return rc;
    float* bin = (float*)(ws + o_bin); unsigned* acc = (unsigned*)(ws + o_acc); uint32_t* list = (uint32_t*)(ws + o_list);
    double* cs = (double*)(ws + o_cs); double* partial = (double*)(ws + o_part); sat_state* st = (sat_state*)(ws + o_st);
    int32_t* cnt = &ctx->d_counters[CNT_TMP];
    BBX_HIP(hipMemcpyAsync(cs, h_cos_sin, (size_t)ntheta * 16, hipMemcpyHostToDevice, s));
    BBX_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(k_sat_init, dim3(1), dim3(128), 0, s, st);
    hipLaunchKernelGGL(k_bin2, dim3((nxb + 255) / 256, nyb), dim3(256), 0, s, d_data, nyb, nxb, bin);
    for (int pass = 0; pass < 4; pass++) {
        hipLaunchKernelGGL(k_clip_pass, dim3(SAT_BLOCKS), dim3(256), 0, s, bin, nb, st, partial);
        hipLaunchKernelGGL(k_clip_update, dim3(1), dim3(256), 0, s, st, partial, SAT_BLOCKS);
    }
    hipLaunchKernelGGL(k_edge_compact, dim3(2048), dim3(256), 0, s, bin, nb, st, list, cnt, (uint32_t)cap, ctx->d_err);
    if ((size_t)nrho * 4 <= 150 * 1024) {
        // rho histogram of one angle fits in LDS (frames up to ~26k binned pixels across)
        hipLaunchKernelGGL(k_hough_lds, dim3(ntheta), dim3(1024), (size_t)nrho * 4, s, list, cnt, (uint32_t)cap, nxb, cs, nrho, st);
    } else {
        return BBX_ERR_ARG;                                  // frame too large for the LDS histogram
    }
    (void)acc;
    hipLaunchKernelGGL(k_trail_decide, dim3(1), dim3(64), 0, s, st, cs, nrho, nyb, nxb);
    hipLaunchKernelGGL(k_trail_profile, dim3(2048), dim3(256), 0, s, bin, nyb, nxb, st);
    hipLaunchKernelGGL(k_trail_strip, dim3(1), dim3(64), 0, s, st);
    hipLaunchKernelGGL(k_trail_mask, dim3(2048), dim3(256), 0, s, d_mask, ny, nx, st);
    hipLaunchKernelGGL(k_sat_info, dim3(1), dim3(64), 0, s, st, d_info);
    BBX_LAUNCH_CHECK();
    return bbx_count_objects(ctx, ny, nx, d_mask, BBX_MASK_SATELLITE, d_nsats, stream);
}
