// bbx_bkg.hip -- 2-D background mesh (zogy.get_back / mini2back as used by BlackBOX:
// buildref.py:2398-2405, 2480-2495).  [EXT: parity unpinned, the
// conventions are the ones written down in oracle/zogy_core.py]
//
//   k_bkg_boxstats   one workgroup per bkg_boxsize x bkg_boxsize box: the usable pixels
//                    (mask == 0, objmask == 0, value != 0) are sorted once in LDS; the
//                    astropy-style clip (centre = median, std about the mean, 3 sigma,
//                    <= 5 iterations) then only narrows an index range of the sorted array.
//                    One read of the frame + mask: 5N bytes, HBM-bound.
//   k_mini_fill_filter  NaN boxes <- nan-median of 3x3 neighbours (repeated), then a 3x3
//                    median filter with replicated edges, on the 176x176 mini image.
//   k_spline_zoom    scipy.ndimage.zoom(order=3, mode='nearest') evaluation: the cubic
//                    B-spline coefficients (tiny, prefiltered on the host) are combined with
//                    per-row / per-column tap weights in float64; optionally fused with the
//                    subtraction from the frame (read 4N + write 4N).
#include "bbx_common.h"
#include "bbx_mednet.h"
#include <rocprim/block/block_radix_sort.hpp>

#define BOX_NS 4096

__global__ __launch_bounds__(256) void k_bkg_boxstats(const float* __restrict__ data, const uint8_t* __restrict__ mask,
                                                      const uint8_t* __restrict__ objmask, int ny, int nx, int box,
                                                      int nbx, float limfrac, float* __restrict__ mini_med,
                                                      float* __restrict__ mini_std) {
    // rocPRIM's block radix sort (256 threads x 16 keys, blocked arrangement) shares its LDS
    // with the sorted array that the clip loop walks afterwards
    using sort_t = rocprim::block_radix_sort<float, 256, BOX_NS / 256>;
    __shared__ union { typename sort_t::storage_type sort; float v[BOX_NS]; } sh;
    float* v = sh.v;
    __shared__ double red[4];
    __shared__ int redi[4];
    __shared__ int s_a, s_b;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int by = blockIdx.x / nbx, bx = blockIdx.x - by * nbx;
    const int npx = box * box;
    int cnt = 0;
    float keys[BOX_NS / 256];
#pragma unroll
    for (int k = 0; k < BOX_NS / 256; k++) {
        const int i = tid * (BOX_NS / 256) + k;
        float val = __builtin_huge_valf();
        if (i < npx) {
            const int y = by * box + i / box, x = bx * box + i % box;
            const size_t o = (size_t)y * nx + x;
            const float d = data[o];
            const bool rej = (mask[o] != 0) || (objmask && objmask[o] != 0) || (d == 0.f) || !(d == d);
            if (!rej) { val = d; cnt++; }
        }
        keys[k] = val;
    }
    cnt = wave_sum_i32(cnt);
    if (lane == 0) redi[wid] = cnt;
    __syncthreads();
    const int n0 = redi[0] + redi[1] + redi[2] + redi[3];
    __syncthreads();
    if ((float)n0 < limfrac * (float)npx || n0 == 0) {
        if (tid == 0) { const float nanv = __uint_as_float(0x7fc00000u); mini_med[blockIdx.x] = nanv; mini_std[blockIdx.x] = nanv; }
        return;
    }
    sort_t().sort(keys, sh.sort);                        // ascending; +inf padding ends up last
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BOX_NS / 256; k++) v[tid * (BOX_NS / 256) + k] = keys[k];
    __syncthreads();
    if (tid == 0) { s_a = 0; s_b = n0; }
    __syncthreads();
    double mean = 0.0, sd = 0.0, med = 0.0;
    for (int it = 0; it <= 5; it++) {
        const int a = s_a, b = s_b, n = b - a;
        // mean and std (about the mean, ddof 0) of the survivors
        double s = 0.0;
        for (int i = a + tid; i < b; i += 256) s += (double)v[i];
        s = wave_sum_f64(s);
        if (lane == 0) red[wid] = s;
        __syncthreads();
        mean = ((red[0] + red[1]) + (red[2] + red[3])) / (double)n;
        __syncthreads();
        double q = 0.0;
        for (int i = a + tid; i < b; i += 256) { const double t = mean - (double)v[i]; q += t * t; }
        q = wave_sum_f64(q);
        if (lane == 0) red[wid] = q;
        __syncthreads();
        sd = sqrt(((red[0] + red[1]) + (red[2] + red[3])) / (double)n);
        med = (n & 1) ? (double)v[a + n / 2] : ((double)v[a + n / 2 - 1] + (double)v[a + n / 2]) * 0.5;
        __syncthreads();
        if (it == 5) break;                               // statistics of the survivors after 5 clips
        const double lo = med - 3.0 * sd, hi = med + 3.0 * sd;
        int nlo = 0, nhi = 0;
        for (int i = a + tid; i < b; i += 256) { const double x = (double)v[i]; nlo += (x < lo); nhi += (x > hi); }
        nlo = wave_sum_i32(nlo); nhi = wave_sum_i32(nhi);
        if (lane == 0) { redi[wid] = nlo; }
        __syncthreads();
        const int tlo = redi[0] + redi[1] + redi[2] + redi[3];
        __syncthreads();
        if (lane == 0) { redi[wid] = nhi; }
        __syncthreads();
        const int thi = redi[0] + redi[1] + redi[2] + redi[3];
        __syncthreads();
        if (tlo == 0 && thi == 0) break;                  // nothing clipped: these are the final statistics
        if (tid == 0) { s_a = a + tlo; s_b = b - thi; }
        __syncthreads();
        if (s_b - s_a <= 0) break;
    }
    if (tid == 0) {
        if (s_b - s_a <= 0) { const float nanv = __uint_as_float(0x7fc00000u); mini_med[blockIdx.x] = nanv; mini_std[blockIdx.x] = nanv; }
        else { mini_med[blockIdx.x] = (float)med; mini_std[blockIdx.x] = (float)sd; }
    }
}

// one workgroup; [mini] is updated in place, [tmp] is scratch of the same size
__global__ __launch_bounds__(1024) void k_mini_fill_filter(float* mini, float* tmp, int nby, int nbx, int32_t* err) {
    __shared__ int nbad, nfixed;
    const int n = nby * nbx;
    float* cur = mini; float* nxt = tmp;
    for (int iter = 0; iter < n + 1; iter++) {
        if (threadIdx.x == 0) { nbad = 0; nfixed = 0; }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            float val = cur[i];
            if (!(val == val)) {
                const int y = i / nbx, x = i - y * nbx;
                float w[9]; int m = 0;
                for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
                    const int yy = y + dy, xx = x + dx;
                    if (yy < 0 || xx < 0 || yy >= nby || xx >= nbx) continue;
                    const float t = cur[yy * nbx + xx];
                    if (t == t) w[m++] = t;
                }
                if (m > 0) {
                    for (int a = 1; a < m; a++) { const float t = w[a]; int b = a - 1; while (b >= 0 && w[b] > t) { w[b + 1] = w[b]; b--; } w[b + 1] = t; }
                    val = (m & 1) ? w[m / 2] : (float)(((double)w[m / 2 - 1] + (double)w[m / 2]) * 0.5);
                    atomicAdd(&nfixed, 1);
                } else atomicAdd(&nbad, 1);
            }
            nxt[i] = val;
        }
        __syncthreads();
        float* t = cur; cur = nxt; nxt = t;
        const int nb = nbad, nf = nfixed;
        __syncthreads();
        if (nb == 0 && nf == 0) break;                    // nothing left to fill
        if (nf == 0) break;                               // all-NaN image: cannot be filled
    }
    // 3x3 median, replicated edges: cur -> nxt
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int y = i / nbx, x = i - y * nbx;
        float w[9]; int m = 0;
        for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
            const int yy = min(max(y + dy, 0), nby - 1), xx = min(max(x + dx, 0), nbx - 1);
            w[m++] = cur[yy * nbx + xx];
        }
        BBX_MED9(w);
        nxt[i] = w[4];
    }
    __syncthreads();
    if (nxt != mini) for (int i = threadIdx.x; i < n; i += blockDim.x) mini[i] = nxt[i];
    (void)err;
}

// One workgroup = 256 columns x ZOOM_ROWS rows: the column taps (index + 4 float64 weights per
// pixel, 36 bytes) are read once per thread and reused down the rows -- with a workgroup per row
// they came through the L2 again for every row, 4.5x the bytes of the image itself.
#define ZOOM_ROWS 16
__global__ __launch_bounds__(256) void k_spline_zoom(int ny, int nx, const double* __restrict__ coef, int cnx,
                                                     const int32_t* __restrict__ fy, const double* __restrict__ wy,
                                                     const int32_t* __restrict__ fx, const double* __restrict__ wx,
                                                     float* data, float* bkg) {
    // the 4 coefficient rows of an output row are first folded with the row weights into a
    // short LDS vector (a 256-pixel span touches only a handful of coefficient columns); each
    // pixel then needs 4 taps instead of 16 strided float64 loads
    __shared__ double rowc[2][512];
    const int X0 = blockIdx.x * blockDim.x, X = X0 + threadIdx.x;
    const int j0 = fx[X0] - 1, j1 = fx[min(X0 + (int)blockDim.x - 1, nx - 1)] + 2;      // fx is non-decreasing
    const int span = j1 - j0 + 1;
    const bool live = X < nx;
    const int ix = live ? fx[X] : 0;
    double wxr[4] = {0, 0, 0, 0};
    if (live) { wxr[0] = wx[X * 4]; wxr[1] = wx[X * 4 + 1]; wxr[2] = wx[X * 4 + 2]; wxr[3] = wx[X * 4 + 3]; }
    const int k = ix - 1 - j0;
    const int Y0 = blockIdx.y * ZOOM_ROWS, Y1 = min(ny, Y0 + ZOOM_ROWS);
    if (span <= 32) {
        // the usual case (zoom factor >> 1: a 256-pixel span touches ~10 coefficient columns): the folded
        // coefficient vectors of all ZOOM_ROWS rows at once, one barrier, then 4 taps per pixel and row
        double* rc = &rowc[0][0];                               // [ZOOM_ROWS][32]
        for (int e = threadIdx.x; e < ZOOM_ROWS * 32; e += blockDim.x) {
            const int r = e >> 5, j = e & 31, Y = Y0 + r;
            if (Y < Y1 && j < span) {
                const double* c0 = coef + (size_t)(fy[Y] - 1) * cnx + (j0 + j);
                rc[e] = ((c0[0] * wy[Y * 4] + c0[cnx] * wy[Y * 4 + 1]) + c0[2 * (size_t)cnx] * wy[Y * 4 + 2]) + c0[3 * (size_t)cnx] * wy[Y * 4 + 3];
            }
        }
        __syncthreads();
        if (!live) return;
        for (int Y = Y0; Y < Y1; Y++) {
            const double* r4 = rc + ((Y - Y0) << 5) + k;
            double t = 0.0;
#pragma unroll
            for (int b = 0; b < 4; b++) t += r4[b] * wxr[b];
            const float v = (float)t;
            const size_t o = (size_t)Y * nx + X;
            if (bkg) bkg[o] = v;
            if (data) data[o] = data[o] - v;
        }
        return;
    }
    for (int Y = Y0; Y < Y1; Y++) {
        const int iy = fy[Y];
        double t = 0.0;
        if (span <= 512) {
            double* rc = rowc[(Y - Y0) & 1];
            const double w0 = wy[Y * 4], w1 = wy[Y * 4 + 1], w2 = wy[Y * 4 + 2], w3 = wy[Y * 4 + 3];
            for (int j = threadIdx.x; j < span; j += blockDim.x) {
                const double* c0 = coef + (size_t)(iy - 1) * cnx + (j0 + j);
                rc[j] = ((c0[0] * w0 + c0[cnx] * w1) + c0[2 * (size_t)cnx] * w2) + c0[3 * (size_t)cnx] * w3;
            }
            __syncthreads();                                     // (the other buffer is free again after the next barrier)
            if (live) {
#pragma unroll
                for (int b = 0; b < 4; b++) t += rc[k + b] * wxr[b];
            }
        } else if (live) {
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const double wa = wy[Y * 4 + a];
                const double* row = coef + (size_t)(iy - 1 + a) * cnx + (ix - 1);
#pragma unroll
                for (int b = 0; b < 4; b++) t += row[b] * (wa * wxr[b]);
            }
        }
        if (live) {
            const float v = (float)t;
            const size_t o = (size_t)Y * nx + X;
            if (bkg) bkg[o] = v;
            if (data) data[o] = data[o] - v;
        }
    }
}

extern "C" {

int bbx_bkg_boxstats(bbx_ctx* ctx, int ny, int nx, int box, const float* d_data, const uint8_t* d_mask,
                     const uint8_t* d_objmask, float limfrac, float* d_mini_med, float* d_mini_std, void* stream) {
    if (!ctx || !d_data || !d_mask || !d_mini_med || !d_mini_std) return BBX_ERR_ARG;
    if (box < 2 || box > 64 || ny % box || nx % box) return BBX_ERR_ARG;
    const int nby = ny / box, nbx = nx / box;
    hipLaunchKernelGGL(k_bkg_boxstats, dim3(nby * nbx), dim3(256), 0, (hipStream_t)stream, d_data, d_mask, d_objmask,
                       ny, nx, box, nbx, limfrac, d_mini_med, d_mini_std);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_mini_fill_filter(bbx_ctx* ctx, int nby, int nbx, float* d_mini, void* stream) {
    if (!ctx || !d_mini || nby < 1 || nbx < 1 || (size_t)nby * nbx > (1u << 22)) return BBX_ERR_ARG;
    int rc;
    float* tmp = (float*)bbx_ws(ctx, WS_MISC, (size_t)nby * nbx * 4 + 256, &rc); if (rc) return rc;
    hipLaunchKernelGGL(k_mini_fill_filter, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_mini, tmp + 64, nby, nbx, ctx->d_err);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_spline_zoom(bbx_ctx* ctx, int ny, int nx, const double* d_coef, int cny, int cnx, const int32_t* d_fy,
                    const double* d_wy, const int32_t* d_fx, const double* d_wx, float* d_data, float* d_bkg,
                    void* stream) {
    if (!ctx || !d_coef || !d_fy || !d_wy || !d_fx || !d_wx || (!d_data && !d_bkg) || ny < 1 || nx < 1 || cny < 4 || cnx < 4)
        return BBX_ERR_ARG;
    hipLaunchKernelGGL(k_spline_zoom, dim3((nx + 255) / 256, (ny + ZOOM_ROWS - 1) / ZOOM_ROWS), dim3(256), 0, (hipStream_t)stream, ny, nx, d_coef, cnx,
                       d_fy, d_wy, d_fx, d_wx, d_data, d_bkg);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

}  // extern "C"
