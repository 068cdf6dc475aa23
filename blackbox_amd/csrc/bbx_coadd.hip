// bbx_coadd.hip -- reference co-add: what buildref.py does per
// input image before SWarp (prep_inputimages 2442-2777), SWarp's LANCZOS3 resampling onto the
// output frame (buildref.py:1727-1763) and its pixel combination (-COMBINE_TYPE, 1733/1815).
//
// All three are HBM-streaming passes:
//   k_coadd_prep     data, background, background-sigma, mask in; data, weights out (float4)
//   k_resample_l3    one output pixel per thread, 6x6 taps gathered through L1/L2 (adjacent
//                    output pixels share 30 of their 36 taps), coordinates from a coarse grid
//   k_combine        one output pixel per thread over the n resampled planes, float64 sums in
//                    image order; order statistics by rank counting in registers (n <= 32)
#include "bbx_common.h"

// ---------------------------------------------------------------------------------
// prep_inputimages: data -= bkg; data[mask == edge] = 0; weights = 1/bkg_std^2 (0 where
// bkg_std == 0 or the mask holds a discarded type)       buildref.py:2602-2624, 2709-2733
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void prep_one(float& d, float b, float s, unsigned m, int discard, int edge, int has_bkg, float& w) {
    if (has_bkg) d = d - b;
    if ((int)m == edge) d = 0.f;
    w = (s != 0.f) ? 1.0f / (s * s) : 0.f;
    if (m & (unsigned)discard) w = 0.f;
}

__global__ __launch_bounds__(256) void k_coadd_prep(float* __restrict__ data, const float* __restrict__ bkg,
                                                    const float* __restrict__ bstd, const uint8_t* __restrict__ mask,
                                                    size_t npix, int discard, int edge, float* __restrict__ wout) {
    const size_t n4 = npix / 4;
    const int has_bkg = bkg != nullptr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 d = ((const float4*)data)[i];
        const float4 s = ((const float4*)bstd)[i];
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_bkg) b = ((const float4*)bkg)[i];
        const uint32_t m = ((const uint32_t*)mask)[i];
        float4 w;
        prep_one(d.x, b.x, s.x, m & 255u, discard, edge, has_bkg, w.x);
        prep_one(d.y, b.y, s.y, (m >> 8) & 255u, discard, edge, has_bkg, w.y);
        prep_one(d.z, b.z, s.z, (m >> 16) & 255u, discard, edge, has_bkg, w.z);
        prep_one(d.w, b.w, s.w, m >> 24, discard, edge, has_bkg, w.w);
        ((float4*)data)[i] = d;
        ((float4*)wout)[i] = w;
    }
    // tail (npix not a multiple of 4)
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        float d = data[i], w;
        prep_one(d, has_bkg ? bkg[i] : 0.f, bstd[i], mask[i], discard, edge, has_bkg, w);
        data[i] = d; wout[i] = w;
    }
}

// ---------------------------------------------------------------------------------
// LANCZOS3 resampling
// ---------------------------------------------------------------------------------
// taps ix-2 .. ix+3 for a position ix + f: k(t) = sinc(t) sinc(t/3), t = f - i, normalised to
// unit sum; float64, rounded to float32 at the end (oracle/coadd.py lanczos3_taps)
__device__ __forceinline__ void l3_taps(double f, float* k) {
    double v[6], sum = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const double t = f - (double)(i - 2);
        double val;
        if (t == 0.0) val = 1.0;
        else if (fabs(t) >= 3.0) val = 0.0;
        else {
            const double a = M_PI * t, b = M_PI * (t / 3.0);
            val = (sin(a) / a) * (sin(b) / b);
        }
        v[i] = val; sum += val;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) k[i] = (float)(v[i] / sum);
}

#define RS_BIG 1e30f
struct rs_args {
    const float* in; const float* win;
    int in_ny, in_nx, out_ny, out_nx;
    const double* grid; int gny, gnx, gstep;
    float fscale;
    float* out; float* wout;
};

__global__ __launch_bounds__(256) void k_resample_l3(rs_args a) {
    const int X = blockIdx.x * 64 + (threadIdx.x & 63);
    const int Y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (X >= a.out_nx || Y >= a.out_ny) return;
    // input position: bilinear interpolation (float64) of the coarse grid
    const int gj = Y / a.gstep, gi = X / a.gstep;
    const double fy = (double)(Y - gj * a.gstep) / (double)a.gstep, fx = (double)(X - gi * a.gstep) / (double)a.gstep;
    const double* g00 = a.grid + ((size_t)gj * a.gnx + gi) * 2;
    const double* g10 = g00 + (size_t)a.gnx * 2;
    double pos[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const double p = g00[k] + (g00[2 + k] - g00[k]) * fx;
        const double q = g10[k] + (g10[2 + k] - g10[k]) * fx;
        pos[k] = p + (q - p) * fy;
    }
    const double xf = floor(pos[0]), yf = floor(pos[1]);
    const size_t o = (size_t)Y * a.out_nx + X;
    // footprint inside the input image?  (also rejects NaN positions)
    if (!(xf - 2.0 >= 0.0 && xf + 3.0 < (double)a.in_nx && yf - 2.0 >= 0.0 && yf + 3.0 < (double)a.in_ny)) {
        a.out[o] = 0.f; a.wout[o] = 0.f;
        return;
    }
    const int ix = (int)xf, iy = (int)yf;
    float kx[6], ky[6];
    l3_taps(pos[0] - xf, kx);
    l3_taps(pos[1] - yf, ky);
    float acc = 0.f, vacc = 0.f;
    bool bad = false;
    const float* p = a.in + (size_t)(iy - 2) * a.in_nx + (ix - 2);
    const float* pw = a.win + (size_t)(iy - 2) * a.in_nx + (ix - 2);
#pragma unroll
    for (int j = 0; j < 6; j++) {
        float row = 0.f, vrow = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const float f = p[i], w = pw[i];
            const float v = (w > 0.f) ? 1.0f / w : RS_BIG;
            bad |= !(w > 0.f);
            row = row + kx[i] * f;
            vrow = vrow + kx[i] * v;
        }
        acc = acc + ky[j] * row;
        vacc = vacc + ky[j] * vrow;
        p += a.in_nx; pw += a.in_nx;
    }
    const float fs = a.fscale;
    const float vout = (fs * fs) * vacc;
    a.out[o] = fs * acc;
    a.wout[o] = (!bad && vout > 0.f) ? 1.0f / vout : 0.f;
}

// ---------------------------------------------------------------------------------
// combination of n resampled planes
// ---------------------------------------------------------------------------------
enum { CB_WEIGHTED = 0, CB_AVERAGE = 1, CB_MEDIAN = 2, CB_CLIPPED = 3, CB_MIN = 4, CB_MAX = 5, CB_SUM = 6 };

struct cb_args {
    const float* cube; const float* wcube;
    long long stride; size_t npix; int n;
    float clip_sigma, clip_ampfrac;
    float* out; float* wout;
    uint8_t* clipmask;               // [n][npix] or null
    unsigned long long* nclip;       // [n] or null
};

// element of 0-based rank r among the valid values (ties by index): rank counting, no
// data-dependent register indexing
template <int CAP>
__device__ __forceinline__ float rank_pick(const float (&f)[CAP], unsigned valid, int n, int r) {
    float res = 0.f;
#pragma unroll
    for (int i = 0; i < CAP; i++) {
        if (i < n && (valid >> i & 1u)) {
            int rk = 0;
#pragma unroll
            for (int j = 0; j < CAP; j++)
                if (j < n && (valid >> j & 1u)) rk += (f[j] < f[i] || (f[j] == f[i] && j < i)) ? 1 : 0;
            if (rk == r) res = f[i];
        }
    }
    return res;
}

template <int TYPE, int CAP>
__global__ __launch_bounds__(256) void k_combine(cb_args a) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < a.npix;
    float f[CAP], w[CAP];
    unsigned valid = 0;
#pragma unroll
    for (int i = 0; i < CAP; i++) {
        f[i] = 0.f; w[i] = 0.f;
        if (live && i < a.n) {
            f[i] = a.cube[(size_t)i * a.stride + p];
            w[i] = a.wcube[(size_t)i * a.stride + p];
            if (w[i] > 0.f) valid |= 1u << i;
        }
    }
    const int m = __popc(valid);
    double out = 0.0, wout = 0.0;
    unsigned drop = 0;
    if (m > 0) {
        double sw = 0.0, swf = 0.0, sf = 0.0, sinv = 0.0;
#pragma unroll
        for (int i = 0; i < CAP; i++)
            if (valid >> i & 1u) {
                const double wd = (double)w[i], fd = (double)f[i];
                sw += wd; swf += wd * fd; sf += fd; sinv += 1.0 / wd;
            }
        if (TYPE == CB_WEIGHTED) { out = swf / sw; wout = sw; }
        else if (TYPE == CB_AVERAGE) { out = sf / (double)m; wout = (double)m * (double)m / sinv; }
        else if (TYPE == CB_SUM) { out = sf; wout = 1.0 / sinv; }
        else if (TYPE == CB_MIN || TYPE == CB_MAX) {
            bool first = true;
#pragma unroll
            for (int i = 0; i < CAP; i++)
                if (valid >> i & 1u) {
                    const bool better = first || (TYPE == CB_MIN ? (double)f[i] < out : (double)f[i] > out);
                    if (better) { out = (double)f[i]; wout = (double)w[i]; first = false; }
                }
        } else {
            const double lo = (double)rank_pick<CAP>(f, valid, a.n, (m - 1) / 2);
            const double hi = (double)rank_pick<CAP>(f, valid, a.n, m / 2);
            const double med = 0.5 * (lo + hi);
            if (TYPE == CB_MEDIAN) { out = med; wout = (2.0 / M_PI) * (double)m * (double)m / sinv; }
            else {
                double sw2 = 0.0, swf2 = 0.0;
                int nk = 0;
#pragma unroll
                for (int i = 0; i < CAP; i++)
                    if (valid >> i & 1u) {
                        const double wd = (double)w[i], fd = (double)f[i];
                        const bool d = fabs(fd - med) > (double)a.clip_sigma * sqrt(1.0 / wd) + (double)a.clip_ampfrac * fabs(med);
                        if (d) drop |= 1u << i;
                        else { sw2 += wd; swf2 += wd * fd; nk++; }
                    }
                if (nk > 0) { out = swf2 / sw2; wout = sw2; }
                else { out = swf / sw; wout = sw; drop = 0; }
            }
        }
    }
    if (live) { a.out[p] = (float)out; a.wout[p] = (float)wout; }
    if (TYPE == CB_CLIPPED) {
#pragma unroll
        for (int i = 0; i < CAP; i++) {
            if (i < a.n) {                                      // uniform
                const bool d = live && (drop >> i & 1u);
                if (a.clipmask && live) a.clipmask[(size_t)i * a.npix + p] = d ? 1 : 0;
                if (a.nclip) {
                    const unsigned long long bal = __ballot(d);
                    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&a.nclip[i], (unsigned long long)__popcll(bal));
                }
            }
        }
    }
}

template <int TYPE>
static void launch_combine(const cb_args& a, hipStream_t s) {
    const dim3 grid((unsigned)((a.npix + 255) / 256)), block(256);
    if (a.n <= 4) hipLaunchKernelGGL((k_combine<TYPE, 4>), grid, block, 0, s, a);
    else if (a.n <= 8) hipLaunchKernelGGL((k_combine<TYPE, 8>), grid, block, 0, s, a);
    else if (a.n <= 16) hipLaunchKernelGGL((k_combine<TYPE, 16>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_combine<TYPE, 32>), grid, block, 0, s, a);
}

extern "C" {

int bbx_coadd_prep(bbx_ctx* ctx, int64_t npix, float* d_data, const float* d_bkg, const float* d_bkg_std,
                   const uint8_t* d_mask, int discard_bits, int edge_value, float* d_weights, void* stream) {
    if (!ctx || !d_data || !d_bkg_std || !d_mask || !d_weights || npix <= 0) return BBX_ERR_ARG;
    if (((uintptr_t)d_data | (uintptr_t)d_bkg_std | (uintptr_t)d_weights | (uintptr_t)d_bkg) % 16 || ((uintptr_t)d_mask) % 4)
        return BBX_ERR_ARG;
    hipLaunchKernelGGL(k_coadd_prep, dim3(4096), dim3(256), 0, (hipStream_t)stream, d_data, d_bkg, d_bkg_std, d_mask,
                       (size_t)npix, discard_bits & 255, edge_value, d_weights);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_resample_lanczos3(bbx_ctx* ctx, int in_ny, int in_nx, const float* d_in, const float* d_win, int out_ny,
                          int out_nx, const double* d_grid, int gny, int gnx, int gstep, float fscale, float* d_out,
                          float* d_wout, void* stream) {
    if (!ctx || !d_in || !d_win || !d_grid || !d_out || !d_wout) return BBX_ERR_ARG;
    if (in_ny < 6 || in_nx < 6 || out_ny < 1 || out_nx < 1 || gstep < 1) return BBX_ERR_ARG;
    // the grid must hold the node after the last pixel in both directions
    if ((out_ny - 1) / gstep + 1 >= gny || (out_nx - 1) / gstep + 1 >= gnx) return BBX_ERR_ARG;
    if ((size_t)in_ny * in_nx >= 0x7fffffffull * 4) return BBX_ERR_ARG;
    rs_args a;
    a.in = d_in; a.win = d_win; a.in_ny = in_ny; a.in_nx = in_nx; a.out_ny = out_ny; a.out_nx = out_nx;
    a.grid = d_grid; a.gny = gny; a.gnx = gnx; a.gstep = gstep; a.fscale = fscale; a.out = d_out; a.wout = d_wout;
    hipLaunchKernelGGL(k_resample_l3, dim3((out_nx + 63) / 64, (out_ny + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

int bbx_coadd_combine(bbx_ctx* ctx, int n, int64_t npix, const float* d_cube, const float* d_wcube, int64_t plane_stride,
                      int combine_type, float clip_sigma, float clip_ampfrac, float* d_out, float* d_wout,
                      uint8_t* d_clipmask, int64_t* d_nclip, void* stream) {
    if (!ctx || !d_cube || !d_wcube || !d_out || !d_wout || n < 1 || n > 32 || npix <= 0 || plane_stride < npix)
        return BBX_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    cb_args a;
    a.cube = d_cube; a.wcube = d_wcube; a.stride = plane_stride; a.npix = (size_t)npix; a.n = n;
    a.clip_sigma = clip_sigma; a.clip_ampfrac = clip_ampfrac; a.out = d_out; a.wout = d_wout;
    a.clipmask = combine_type == CB_CLIPPED ? d_clipmask : nullptr;
    a.nclip = combine_type == CB_CLIPPED ? (unsigned long long*)d_nclip : nullptr;
    if (d_nclip) BBX_HIP(hipMemsetAsync(d_nclip, 0, (size_t)n * sizeof(int64_t), s));
    switch (combine_type) {
        case CB_WEIGHTED: launch_combine<CB_WEIGHTED>(a, s); break;
        case CB_AVERAGE: launch_combine<CB_AVERAGE>(a, s); break;
        case CB_MEDIAN: launch_combine<CB_MEDIAN>(a, s); break;
        case CB_CLIPPED: launch_combine<CB_CLIPPED>(a, s); break;
        case CB_MIN: launch_combine<CB_MIN>(a, s); break;
        case CB_MAX: launch_combine<CB_MAX>(a, s); break;
        case CB_SUM: launch_combine<CB_SUM>(a, s); break;
        default: return BBX_ERR_ARG;
    }
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}

}  // extern "C"
