// bbx_stack.hip -- master calibration frames (reference master_prep, blackbox.py:4906-5073)
//
//   master_cube[i] /= MEDSEC_i            (flats only, float32 division; 4929-4941)
//   master_median = np.median(master_cube, axis=0)                          (4984)
//   flat: pixels that are BPM-edge or <= 0 -> 1                           (5071-5073)
//
// Per pixel the <= 32 values live in registers and go through an odd-even transposition
// network; even counts return the float32 mean of the two middle values like np.median.
// Pure streaming: nframes * 4N bytes read, 4N written -> HBM-bound (20 x 446 MB = 8.9 GB
// per master).  NaN inputs are not propagated like numpy does (min/max network).
#include "bbx_common.h"

#define STACK_MAX 32
struct stack_args {
    const float* f[STACK_MAX];
    float norm[STACK_MAX];
    const uint8_t* bpm; float* out; size_t n4; int flat_fix;
};

template <int N>
__global__ __launch_bounds__(256) void k_median_stack(stack_args a) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n4; i += (size_t)gridDim.x * blockDim.x) {
        float v[N][4];
#pragma unroll
        for (int k = 0; k < N; k++) {
            const float4 t = *(const float4*)(a.f[k] + i * 4);
            const float d = a.norm[k];
            v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w;
            if (d != 0.f && d != 1.f) {
#pragma unroll
                for (int q = 0; q < 4; q++) v[k][q] = v[k][q] / d;
            }
        }
#pragma unroll
        for (int r = 0; r < N; r++) {
#pragma unroll
            for (int k = (r & 1); k + 1 < N; k += 2) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float lo = fminf(v[k][q], v[k + 1][q]), hi = fmaxf(v[k][q], v[k + 1][q]);
                    v[k][q] = lo; v[k + 1][q] = hi;
                }
            }
        }
        float m[4];
#pragma unroll
        for (int q = 0; q < 4; q++) m[q] = (N & 1) ? v[N / 2][q] : (v[(N - 1) / 2][q] + v[N / 2][q]) * 0.5f;
        if (a.flat_fix) {
            uchar4 b = make_uchar4(0, 0, 0, 0);
            if (a.bpm) b = *(const uchar4*)(a.bpm + i * 4);
            const uint8_t bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int q = 0; q < 4; q++) if (bb[q] == BBX_MASK_EDGE || m[q] <= 0.f) m[q] = 1.f;
        }
        *(float4*)(a.out + i * 4) = make_float4(m[0], m[1], m[2], m[3]);
    }
}

template <int N>
static void launch_stack(const stack_args& a, hipStream_t s) {
    unsigned grid = (unsigned)((a.n4 + 255) / 256);
    if (grid > 256u * 8u) grid = 256u * 8u;
    hipLaunchKernelGGL(k_median_stack<N>, dim3(grid), dim3(256), 0, s, a);
}

extern "C" int bbx_median_stack(bbx_ctx* ctx, int64_t npix, int nframes, const float* const* h_frames,
                                const float* h_norm, const uint8_t* d_bpm, int flat_fix, float* d_out,
                                void* stream) {
    if (!ctx || !h_frames || !d_out || npix <= 0 || nframes < 1 || nframes > STACK_MAX) return BBX_ERR_ARG;
    if (npix % 4 || ((uintptr_t)d_out) % 16 || (d_bpm && ((uintptr_t)d_bpm) % 4)) return BBX_ERR_ARG;
    stack_args a;
    for (int k = 0; k < STACK_MAX; k++) { a.f[k] = nullptr; a.norm[k] = 1.f; }
    for (int k = 0; k < nframes; k++) {
        if (!h_frames[k] || ((uintptr_t)h_frames[k]) % 16) return BBX_ERR_ARG;
        a.f[k] = h_frames[k];
        a.norm[k] = h_norm ? h_norm[k] : 1.f;
    }
    a.bpm = d_bpm; a.out = d_out; a.n4 = (size_t)npix / 4; a.flat_fix = flat_fix;
    hipStream_t s = (hipStream_t)stream;
    switch (nframes) {
#define CASE(N) case N: launch_stack<N>(a, s); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12)
        CASE(13) CASE(14) CASE(15) CASE(16) CASE(17) CASE(18) CASE(19) CASE(20) CASE(21) CASE(22) CASE(23)
        CASE(24) CASE(25) CASE(26) CASE(27) CASE(28) CASE(29) CASE(30) CASE(31) CASE(32)
#undef CASE
    }
    BBX_LAUNCH_CHECK();
    return BBX_OK;
}
