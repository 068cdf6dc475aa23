// bbx_xtalk.hip -- crosstalk correction (reference xtalk_corr, blackbox.py:7138-7258)
//
// The reference stacks the 16 channels, builds a y-flipped copy, and runs four
// float64 matmuls with K = 8.  Here one thread owns one pixel position (y, x) of
// the channel grid: it reads the 16 source values that couple to each other -- the
// 8 lower-row channels at row y and the 8 upper-row channels at the mirrored row
// ysize-1-y -- keeps them in float64 registers, and then takes the 16 victims in turn:
// the victim's 16 coefficients arrive by scalar loads, its own value comes from the registers
// and the corrected value written back in place.  Every pixel is read once and written
// once: 4N + N (mask) + 4N bytes.  (Round 4, SQ counters: the kernel fills 0.86 of its vector issue slots -- the 16 float64
// FMAs per output pixel at 8 cycles each and what surrounds them bound it, not the 3.9 TB/s it moves.)
#include "bbx_common.h"
#include <stdlib.h>

template <int VEC> __device__ __forceinline__ void xtalk_fetch(const float* data, const uint8_t* mask, size_t off, float (&v)[VEC], uint8_t (&m)[VEC]) {
    if (VEC == 4) {
        const float4 f = *(const float4*)(data + off);
        const uchar4 b = *(const uchar4*)(mask + off);
        v[0] = f.x; v[1 % VEC] = f.y; v[2 % VEC] = f.z; v[3 % VEC] = f.w;
        m[0] = b.x; m[1 % VEC] = b.y; m[2 % VEC] = b.z; m[3 % VEC] = b.w;
    } else if (VEC == 2) {
        const float2 f = *(const float2*)(data + off);
        const uchar2 b = *(const uchar2*)(mask + off);
        v[0] = f.x; v[VEC - 1] = f.y; m[0] = b.x; m[VEC - 1] = b.y;
    } else {
        v[0] = data[off]; m[0] = mask[off];
    }
}

// VEC pixels per thread and channel (float4 / uchar4 accesses when VEC == 4).
// [cf] is the transposed matrix: cf.v[v * 16 + s] = coefficient of source channel s on victim v.
// The loop over the victims is a real loop: the 16 coefficients of a victim are fetched by scalar
// loads from the kernel-argument segment when its turn comes.  (Unrolled, the 256 coefficients
// are loop invariants that the compiler keeps in 512 SGPRs it does not have: it spilled them to
// VGPR lanes and the kernel spent its time in v_readlane, 0.72 ms per frame.)
// Round 3: the 16 values stay in registers as the float32 they were read as, with two bit sets (usable as a source /
// edge pixel as a victim); the float64 source terms are converted when a victim needs them.  Until then the kernel kept
// 16 float64 sources and read every victim's value and mask byte a second time -- from L2 in theory, but the PMC
// traffic was 1.54 GB for 1.0 GB of algorithmic bytes.
template <int VEC>
__global__ __launch_bounds__(256) void k_xtalk(float* data, const uint8_t* __restrict__ mask, bbx_dims d, f64x256 cf) {
    const int ngx = d.xsz / VEC;
    const size_t total = (size_t)d.ysz * ngx;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(t / ngx), x = (int)(t - (size_t)y * ngx) * VEC;
        const size_t off_lo = (size_t)y * d.nx + x, off_hi = (size_t)(d.ysz + (d.ysz - 1 - y)) * d.nx + x;
        float val[16][VEC];
        unsigned use[VEC], edge[VEC];
#pragma unroll
        for (int q = 0; q < VEC; q++) use[q] = edge[q] = 0u;
#pragma unroll
        for (int c = 0; c < 16; c++) {
            const size_t off = ((c >> 3) ? off_hi : off_lo) + (size_t)(c & 7) * d.xsz;
            uint8_t m[VEC];
            xtalk_fetch<VEC>(data, mask, off, val[c], m);
#pragma unroll
            for (int q = 0; q < VEC; q++) {
                // mask_source: positive, not bad, not cosmic (7178-7180); mask_victim: not edge (7184)
                if ((val[c][q] > 0.f) && !(m[q] & BBX_MASK_BAD) && !(m[q] & BBX_MASK_COSMIC)) use[q] |= 1u << c;
                if (m[q] & BBX_MASK_EDGE) edge[q] |= 1u << c;
            }
        }
#ifndef XTALK_UNROLL
#define XTALK_UNROLL 2                     // (1: 193 us, 2: 188, 4: 191 -- the next victim's coefficients load while this one computes)
#endif
#pragma unroll XTALK_UNROLL
        for (int v = 0; v < 16; v++) {
            const size_t off = ((v >> 3) ? off_hi : off_lo) + (size_t)(v & 7) * d.xsz;
            const double* cv = &cf.v[v * 16];
            float o[VEC];
#pragma unroll
            for (int q = 0; q < VEC; q++) {
                double q_lo = 0.0, q_hi = 0.0;                 // the two K=8 quadrant products
#pragma unroll
                for (int s = 0; s < 8; s++) q_lo = fma(((use[q] >> s) & 1u) ? (double)val[s][q] : 0.0, cv[s], q_lo);
#pragma unroll
                for (int s = 8; s < 16; s++) q_hi = fma(((use[q] >> s) & 1u) ? (double)val[s][q] : 0.0, cv[s], q_hi);
                const double corr = (0.0 + q_lo) + q_hi;
                // the victim's own value: register v of the 16 (a run-time index: picked by a chain of selects)
                // the victim's own value: register v of the 16, a run-time but wave-uniform index -- the compiler reads it in
                // GPR-index mode (s_set_gpr_idx_on), one move instead of the chain of 15 selects of rounds 2-3 (a quarter of
                // the kernel's vector instructions; the kernel is issue-bound: 232 -> 198 us)
                const float own = val[v][q];
                o[q] = (float)((double)own - (!((edge[q] >> v) & 1u) ? corr : corr * 0.0));
            }
            if (VEC == 4) *(float4*)(data + off) = make_float4(o[0], o[1 % VEC], o[2 % VEC], o[3 % VEC]);
            else if (VEC == 2) *(float2*)(data + off) = make_float2(o[0], o[VEC - 1]);
            else data[off] = o[0];
        }
    }
}

extern "C" int bbx_xtalk(bbx_ctx* ctx, const bbx_geom* g, float* d_data, const uint8_t* d_mask,
                         const double* h_coeffs, void* stream) {
    if (!ctx || !d_data || !d_mask || !h_coeffs) return BBX_ERR_ARG;
    bbx_dims d; int rc = bbx_make_dims(g, &d); if (rc) return rc;
    f64x256 cf;                                             // transposed: the coefficients of a victim are contiguous
    for (int sc = 0; sc < 16; sc++) for (int v = 0; v < 16; v++) cf.v[v * 16 + sc] = h_coeffs[sc * 16 + v];
    // measured on MI355X (full frame): one pixel per thread 0.26 ms, two 0.40, four 0.32 -- the
    // narrow variant needs 52 VGPRs and runs 8 waves per SIMD, which hides the 16 channel streams'
    // latency best.  BBX_XTALK_VEC=2|4 selects the wider ones where geometry and pointers allow.
    int vw = 1;
    if (getenv("BBX_XTALK_VEC")) {
        const int w = atoi(getenv("BBX_XTALK_VEC"));
        if (w == 2 && d.xsz % 2 == 0 && d.nx % 2 == 0 && ((uintptr_t)d_data) % 8 == 0 && ((uintptr_t)d_mask) % 2 == 0) vw = 2;
        if (w == 4 && d.xsz % 4 == 0 && d.nx % 4 == 0 && ((uintptr_t)d_data) % 16 == 0 && ((uintptr_t)d_mask) % 4 == 0) vw = 4;
    }
    const size_t total = (size_t)d.ysz * (d.xsz / vw);
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 256u * 16u) grid = 256u * 16u;                // 2k ... 1M workgroups measure the same
    bbx_prof_start(ctx, BBX_PROF_XTALK, (hipStream_t)stream);
    if (vw == 4) hipLaunchKernelGGL(k_xtalk<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_data, d_mask, d, cf);
    else if (vw == 2) hipLaunchKernelGGL(k_xtalk<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_data, d_mask, d, cf);
    else hipLaunchKernelGGL(k_xtalk<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, d_data, d_mask, d, cf);
    bbx_prof_stop(ctx, (hipStream_t)stream);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}
