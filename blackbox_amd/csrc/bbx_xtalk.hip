// bbx_xtalk.hip -- crosstalk correction (reference xtalk_corr, blackbox.py:7138-7258)
//
// The reference stacks the 16 channels, builds a y-flipped copy, and runs four
// float64 matmuls with K = 8.  Here one thread owns one pixel position (y, x) of
// the channel grid: it reads the 16 source values that couple to each other -- the
// 8 lower-row channels at row y and the 8 upper-row channels at the mirrored row
// ysize-1-y -- applies the 16x16 coefficient matrix in float64 registers and writes
// the 16 corrected victims back in place.  Every pixel is read once and written
// once: 4N + N (mask) + 4N bytes, HBM-bound (16 float64 FMAs per output pixel are
// far below the vector rate, so no MFMA reshaping).
#include "bbx_common.h"
#include <stdlib.h>

// VEC pixels per thread and channel (4 when xsize_chan % 4 == 0: float4 / uchar4 accesses)
template <int VEC>
__global__ __launch_bounds__(256) void k_xtalk(float* data, const uint8_t* __restrict__ mask, bbx_dims d, f64x256 cf) {
    const int ngx = d.xsz / VEC;
    const size_t total = (size_t)d.ysz * ngx;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(t / ngx), x = (int)(t - (size_t)y * ngx) * VEC;
        size_t off[16];
        double src[16][VEC];
        float val[16][VEC];
        bool victim_ok[16][VEC];
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const int iy = c >> 3, ix = c & 7;
            const int Y = (iy == 0) ? y : (d.ysz + (d.ysz - 1 - y));
            off[c] = (size_t)Y * d.nx + (size_t)ix * d.xsz + x;
            float v[VEC]; uint8_t m[VEC];
            if (VEC == 4) {
                const float4 f = *(const float4*)(data + off[c]);
                const uchar4 b = *(const uchar4*)(mask + off[c]);
                v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
                m[0] = b.x; m[1] = b.y; m[2] = b.z; m[3] = b.w;
            } else if (VEC == 2) {
                const float2 f = *(const float2*)(data + off[c]);
                const uchar2 b = *(const uchar2*)(mask + off[c]);
                v[0] = f.x; v[1] = f.y; m[0] = b.x; m[1] = b.y;
            } else {
                v[0] = data[off[c]]; m[0] = mask[off[c]];
            }
#pragma unroll
            for (int q = 0; q < VEC; q++) {
                val[c][q] = v[q];
                // mask_source: positive, not bad, not cosmic (7178-7180); mask_victim: not edge (7184)
                const bool use = (v[q] > 0.f) && !(m[q] & BBX_MASK_BAD) && !(m[q] & BBX_MASK_COSMIC);
                src[c][q] = use ? (double)v[q] : 0.0;
                victim_ok[c][q] = !(m[q] & BBX_MASK_EDGE);
            }
        }
#pragma unroll
        for (int v = 0; v < 16; v++) {
            float o[VEC];
#pragma unroll
            for (int q = 0; q < VEC; q++) {
                double q_lo = 0.0, q_hi = 0.0;                 // the two K=8 quadrant products
#pragma unroll
                for (int s = 0; s < 8; s++) q_lo = fma(src[s][q], cf.v[s * 16 + v], q_lo);
#pragma unroll
                for (int s = 8; s < 16; s++) q_hi = fma(src[s][q], cf.v[s * 16 + v], q_hi);
                const double corr = (0.0 + q_lo) + q_hi;
                o[q] = (float)((double)val[v][q] - (victim_ok[v][q] ? corr : corr * 0.0));
            }
            if (VEC == 4) *(float4*)(data + off[v]) = make_float4(o[0], o[1], o[2], o[3]);
            else if (VEC == 2) *(float2*)(data + off[v]) = make_float2(o[0], o[VEC - 1]);
            else data[off[v]] = o[0];
        }
    }
}

extern "C" int bbx_xtalk(bbx_ctx* ctx, const bbx_geom* g, float* d_data, const uint8_t* d_mask,
                         const double* h_coeffs, void* stream) {
    if (!ctx || !d_data || !d_mask || !h_coeffs) return BBX_ERR_ARG;
    bbx_dims d; int rc = bbx_make_dims(g, &d); if (rc) return rc;
    f64x256 cf;
    for (int i = 0; i < 256; i++) cf.v[i] = h_coeffs[i];
    // measured on MI355X: the 4-wide variant is register-bound and slower than one pixel per thread;
    // two pixels per thread (8-byte loads, 2-byte mask loads) is the fastest of the three
    const int vw = (getenv("BBX_XTALK_VEC") ? atoi(getenv("BBX_XTALK_VEC")) : 2);
    const bool vec2 = vw == 2 && d.xsz % 2 == 0 && ((uintptr_t)d_data) % 8 == 0 && ((uintptr_t)d_mask) % 2 == 0 && d.nx % 2 == 0;
    const bool vec = vw == 4 && d.xsz % 4 == 0 && d.nx % 4 == 0;
    const size_t total = (size_t)d.ysz * (d.xsz / (vec ? 4 : vec2 ? 2 : 1));
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 256u * 16u) grid = 256u * 16u;
    bbx_prof_start(ctx, BBX_PROF_XTALK, (hipStream_t)stream);
    if (vec) hipLaunchKernelGGL(k_xtalk<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_data, d_mask, d, cf);
    else if (vec2) hipLaunchKernelGGL(k_xtalk<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_data, d_mask, d, cf);
    else hipLaunchKernelGGL(k_xtalk<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_data, d_mask, d, cf);
    bbx_prof_stop(ctx, (hipStream_t)stream);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}
