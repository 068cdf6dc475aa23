// bbx_psf.hip -- PSFEx model evaluation: the PSF of every
// source is a polynomial in its (normalised) position over a cube of basis images,
//     stamp[s][p] = sum_k  term[s][k] * basis[k][p],   term = x'^i y'^j, i + j <= poldeg
// i.e. the dense contraction [n_src x n_coeff] . [n_coeff x stamp_px].  n_coeff is 6-15, so the
// kernel is bound by the write of the stamps; it runs on the f32 MFMA (v_mfma_f32_32x32x2_f32)
// because that leaves the vector unit idle and needs one register per operand.  Numerics of
// that instruction: a k-ordered float32 fmaf chain, one rounding per product-add
// (cdna_hip_programming.md, "FP32-input MFMA"), which oracle/zogy_core.psf_model reproduces.
#include "bbx_common.h"

typedef float f32x16v __attribute__((ext_vector_type(16)));

#define PSF_NT 4            // 32-column accumulator tiles per wave -> 32 x 128 outputs per wave

// wave tile: sources [m0, m0+32) x stamp pixels [n0, n0+128)
__global__ __launch_bounds__(256) void k_psf_model(int nsrc, int ncoef, int npix, const float* __restrict__ terms,
                                                   const float* __restrict__ basis, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = (blockIdx.y * 4 + wave) * 32, n0 = blockIdx.x * (32 * PSF_NT);
    if (m0 >= nsrc) return;                                      // wave-uniform
    const int r = lane & 31, kh = lane >> 5;
    f32x16v acc[PSF_NT];
#pragma unroll
    for (int t = 0; t < PSF_NT; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
    const int row = m0 + r;
    for (int kk = 0; kk < ncoef; kk += 2) {
        const int k = kk + kh;
        // A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31]; out-of-range -> 0
        const float a = (row < nsrc && k < ncoef) ? terms[(size_t)row * ncoef + k] : 0.f;
#pragma unroll
        for (int t = 0; t < PSF_NT; t++) {
            const int col = n0 + 32 * t + r;
            const float b = (col < npix && k < ncoef) ? basis[(size_t)k * npix + col] : 0.f;
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
    }
    // C/D: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int t = 0; t < PSF_NT; t++) {
        const int col = n0 + 32 * t + r;
        if (col >= npix) continue;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int orow = m0 + (i & 3) + 8 * (i >> 2) + 4 * kh;
            if (orow < nsrc) out[(size_t)orow * npix + col] = acc[t][i];
        }
    }
}

extern "C" int bbx_psf_model(bbx_ctx* ctx, int nsrc, int ncoef, int npix, const float* d_terms, const float* d_basis,
                             float* d_out, void* stream) {
    if (!ctx || !d_terms || !d_basis || !d_out || nsrc < 1 || ncoef < 1 || ncoef > 64 || npix < 1) return BBX_ERR_ARG;
    const dim3 grid((npix + 32 * PSF_NT - 1) / (32 * PSF_NT), (nsrc + 127) / 128);
    hipLaunchKernelGGL(k_psf_model, grid, dim3(256), 0, (hipStream_t)stream, nsrc, ncoef, npix, d_terms, d_basis, d_out);
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}
