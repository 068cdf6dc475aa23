// bbx_clipstats.hip -- sigma-clipped statistics of ONE large sample (bbx_frame_clipped_stats): the header statistics
// optimal_subtraction takes from the Scorr and Fpsferr frames (Z-SCMED / Z-SCSTD, Z-FPEMED / Z-FPESTD).
//
// astropy.stats.sigma_clipped_stats(x, sigma, maxiters, cenfunc = median, stdfunc = std) keeps, round after round, the
// values inside median +- sigma * std of the previous survivors: the survivors are always ONE RUN of the sorted sample.
// So the sample is sorted once (radix sort, rocPRIM through hipCUB) and every round is index arithmetic on the sorted
// array: the median is the middle element (exact, as the bracketed select of bbx_rect_clipped_stats gives it), the sums
// come from per-block float64 sums plus the two ragged ends, the new window from two searches.  ~9 launches
// and ~0.1 ms for the 1.7 10^6 lattice points of a 10560^2 frame; the generic path (segments of a frame, no sort)
// takes 12 launches PER ROUND.
#include "bbx_common.h"
#include <hipcub/hipcub.hpp>

#define CS_BLK 2048                 // sorted values per block sum
#define CS_T 1024                   // threads of the clip kernel

// lattice point i -> its value, or +inf when it does not take part (mask bits other than the cosmic-ray flag, not
// finite (astropy masks those), the masked value 0 on request).  No counter: the values that take part are the ones
// below +inf of the sorted array (one counter for 27 000 waves is a queue of same-address atomics: 0.3 ms).
__global__ __launch_bounds__(256) void k_cs_keys(const float* __restrict__ img, const uint8_t* __restrict__ mask, int nx, int step, int my, int mx,
                                                 int skip_zero, float* __restrict__ keys) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= my * mx) return;
    const int yy = i / mx, xx = i - yy * mx;
    const size_t o = (size_t)yy * step * nx + (size_t)xx * step;
    const float v = img[o];
    const bool ok = fabsf(v) < __builtin_huge_valf() && !(skip_zero && v == 0.f) && !(mask && (mask[o] & ~BBX_MASK_COSMIC));
    keys[i] = ok ? v : __builtin_huge_valf();
}

// float64 sums of x and x^2 over blocks of CS_BLK sorted values (fixed order: the same bits every run)
__global__ __launch_bounds__(256) void k_cs_blocksums(const float* __restrict__ sorted, int n, double* __restrict__ bs) {
    const int b0 = blockIdx.x * CS_BLK;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < CS_BLK / 256; k++) {
        const int i = b0 + k * 256 + (int)threadIdx.x;
        if (i < n) { const float v = sorted[i]; if (v < __builtin_huge_valf()) { const double d = (double)v; s1 += d; s2 += d * d; } }
    }
    __shared__ double sh[2][4];
    s1 = wave_sum_f64(s1); s2 = wave_sum_f64(s2);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s1; sh[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x < 2) bs[2 * blockIdx.x + threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

// workgroup-wide sums, the same on every thread
__device__ __forceinline__ double cs_wg_sum(double v, double* sh) {
    v = wave_sum_f64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < CS_T / 64; w++) t += sh[w];
    return t;
}
__device__ __forceinline__ int cs_wg_count(bool p, int* shi) {
    const int v = wave_sum_i32(p ? 1 : 0);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) shi[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
    for (int w = 0; w < CS_T / 64; w++) t += shi[w];
    return t;
}
// first index in [a, b) of the sorted array whose value is >= w (strict: > w), b if there is none.  A search in steps of
// (b - a) / 1024 by the whole workgroup: three dependent loads for 1.7 10^6 values (a bisection by one thread is 21).
__device__ __forceinline__ int cs_search(const float* __restrict__ s, int a, int b, double w, bool strict, int* shi) {
    while (a < b) {
        const int len = b - a, stride = (len + CS_T - 1) / CS_T;
        const long long i = (long long)a + (long long)threadIdx.x * stride;
        bool below = false;                                        // the probe lies before the boundary
        if (i < b) { const double d = (double)s[i]; below = strict ? !(d > w) : !(d >= w); }
        const int nb = cs_wg_count(below, shi);                    // probes 0 .. nb - 1 are below (the array is sorted)
        if (nb == 0) return a;
        const int last = a + (nb - 1) * stride;                    // the last probe below the boundary
        if (stride == 1) return last + 1;
        a = last + 1;
        b = min(b, last + stride);                                 // the next probe (not below), or the end
    }
    return a;
}

// all rounds in one workgroup.  out[8] = {n, median, mean, sigma (ddof 0), 0, 0, 0, 0}
__global__ __launch_bounds__(CS_T) void k_cs_clip(const float* __restrict__ s, int ntot, const double* __restrict__ bs, double sigma,
                                                  int maxiters, double* __restrict__ out) {
    __shared__ double sh[CS_T / 64];
    __shared__ int shi[CS_T / 64];
    int i0 = 0, i1 = cs_search(s, 0, ntot, (double)__builtin_huge_valf(), false, shi);      // the values that take part
    double wlo = -__builtin_huge_val(), whi = __builtin_huge_val();
    float med = 0.f;
    double n = 0.0, mean = 0.0, sd = 0.0;
    for (int it = 0; it <= maxiters; it++) {
        const int cnt = i1 - i0;
        n = (double)cnt;
        if (cnt <= 0) break;
        const float lo = s[i0 + (cnt - 1) / 2], hi = s[i0 + cnt / 2];
        med = (cnt & 1) ? lo : (lo + hi) * 0.5f;
        // sums over [i0, i1): whole blocks from the block sums, the ragged ends value by value
        const int b0 = (i0 + CS_BLK - 1) / CS_BLK, b1 = i1 / CS_BLK;          // whole blocks b0 .. b1 - 1
        double s1 = 0.0, s2 = 0.0;
        if (b0 < b1) {
            for (int b = b0 + (int)threadIdx.x; b < b1; b += CS_T) { s1 += bs[2 * b]; s2 += bs[2 * b + 1]; }
            for (int i = i0 + (int)threadIdx.x; i < b0 * CS_BLK; i += CS_T) { const double d = (double)s[i]; s1 += d; s2 += d * d; }
            for (int i = b1 * CS_BLK + (int)threadIdx.x; i < i1; i += CS_T) { const double d = (double)s[i]; s1 += d; s2 += d * d; }
        } else {
            for (int i = i0 + (int)threadIdx.x; i < i1; i += CS_T) { const double d = (double)s[i]; s1 += d; s2 += d * d; }
        }
        s1 = cs_wg_sum(s1, sh); s2 = cs_wg_sum(s2, sh);
        mean = s1 / n;
        double var = s2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        sd = sqrt(var);
        if (it == maxiters) break;
        // next round: the survivors inside median +- sigma * std (the window only ever shrinks)
        const double lo2 = (double)med - sigma * sd, hi2 = (double)med + sigma * sd;
        if (lo2 > wlo) wlo = lo2;
        if (hi2 < whi) whi = hi2;
        const int j0 = cs_search(s, i0, i1, wlo, false, shi), j1 = cs_search(s, j0, i1, whi, true, shi);
        if (j0 == i0 && j1 == i1) break;                          // nothing clipped: the remaining rounds change nothing
        i0 = j0; i1 = j1;
    }
    if (threadIdx.x == 0) {
        out[0] = n; out[1] = n > 0 ? (double)med : __longlong_as_double(0x7ff8000000000000LL);
        out[2] = mean; out[3] = sd; out[4] = 0.0; out[5] = 0.0; out[6] = 0.0; out[7] = 0.0;
    }
}

extern "C" int bbx_frame_clipped_stats(bbx_ctx* ctx, int ny, int nx, const float* d_img, const uint8_t* d_mask, int step, double sigma,
                                       int maxiters, int skip_zero, double* d_out, void* stream) {
    if (!ctx || !d_img || !d_out || ny < 1 || nx < 1 || step < 1 || maxiters < 0 || maxiters > 20 || !(sigma > 0.0)) return BBX_ERR_ARG;
    const int my = (ny + step - 1) / step, mx = (nx + step - 1) / step;
    if ((long long)my * mx > (1ll << 30)) return BBX_ERR_ARG;
    const int n = my * mx, nblk = (n + CS_BLK - 1) / CS_BLK;
    hipStream_t s = (hipStream_t)stream;
    size_t tmp = 0;
    BBX_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp, (const float*)nullptr, (float*)nullptr, n, 0, 32, s));
    const size_t o_keys = 256, o_sorted = o_keys + (((size_t)n * 4 + 255) & ~(size_t)255), o_bs = o_sorted + (((size_t)n * 4 + 255) & ~(size_t)255),
                 o_tmp = o_bs + (((size_t)nblk * 16 + 255) & ~(size_t)255);
    int rc;
    char* ws = (char*)bbx_ws(ctx, WS_SEL, o_tmp + tmp + 256, &rc); if (rc) return rc;
    float *keys = (float*)(ws + o_keys), *sorted = (float*)(ws + o_sorted);
    double* bs = (double*)(ws + o_bs);
    hipLaunchKernelGGL(k_cs_keys, dim3((n + 255) / 256), dim3(256), 0, s, d_img, d_mask, nx, step, my, mx, skip_zero ? 1 : 0, keys);
    BBX_HIP(hipcub::DeviceRadixSort::SortKeys(ws + o_tmp, tmp, (const float*)keys, sorted, n, 0, 32, s));
    hipLaunchKernelGGL(k_cs_blocksums, dim3(nblk), dim3(256), 0, s, sorted, n, bs);
    hipLaunchKernelGGL(k_cs_clip, dim3(1), dim3(CS_T), 0, s, sorted, n, bs, sigma, maxiters, d_out);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}
