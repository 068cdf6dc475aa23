// Memory-system probe for the T / U tile layouts of bbx_zogy_frame: what does HBM give a kernel whose workgroups read a
// 505 MB array as pieces of P bytes at a stride of 22.5 KB (the column kernels read 128-byte tiles of 350 row blocks) and
// write their share back contiguously, against the same bytes read contiguously?  768 threads, 2 workgroups per CU by LDS
// (64 KB), the loads of a workgroup all in flight before its stores, like the kernels' load -> store phases without the
// transforms.   hipcc --offload-arch=gfx950 -O3 tools/exp/tile_bw.hip -o gpurun_out/tile_bw && gpurun_out/tile_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// array of nb row blocks x G groups x P bytes ([yb][g][P]): workgroup (g, sub) reads its piece of every row block
template <int P16>   // piece size in 16-byte units (8 = 128 B)
__global__ __launch_bounds__(768) void k_pieces(const float4* __restrict__ in, float4* __restrict__ out, int nb, int G, int nsub, int xcd) {
    extern __shared__ float4 lds[];
    int task = blockIdx.x;
    if (xcd) task = (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8);
    if (task >= G * nsub) return;
    const int g = task % G, sub = task / G;
    const size_t unit = (size_t)nb * G * P16;
    const float4* src = in + (size_t)sub * unit + (size_t)g * P16;
    const int NV = nb * P16;
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int e = threadIdx.x + i * 768;
        if (e < NV) { const int yb = e / P16, j = e - yb * P16; v[i] = src[(size_t)yb * G * P16 + j]; }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { const int e = threadIdx.x + i * 768; if (e < NV) lds[e] = v[i]; }
    __syncthreads();
    float4* dst = out + (size_t)sub * unit + (size_t)g * NV;                  // contiguous share
    for (int e = threadIdx.x; e < NV; e += 768) dst[e] = lds[(e * 7) % NV];   // (any permutation: the stores are what matters)
}

// the same, persistent: a workgroup's loads of the NEXT task are issued before the stores of the current one (registers),
// grid = 2 workgroups per CU
template <int P16>
__global__ __launch_bounds__(768) void k_pieces_persist(const float4* __restrict__ in, float4* __restrict__ out, int nb, int G, int nsub) {
    extern __shared__ float4 lds[];
    const int ntask = G * nsub, per = (ntask + 7) / 8;
    const int band = blockIdx.x % 8, nwg = gridDim.x / 8, k0 = blockIdx.x / 8;
    const size_t unit = (size_t)nb * G * P16;
    const int NV = nb * P16;
    float4 v[4];
    auto load = [&](int task) {
        const int g = task % G, sub = task / G;
        const float4* src = in + (size_t)sub * unit + (size_t)g * P16;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int e = min((int)threadIdx.x + i * 768, NV - 1);
            const int yb = e / P16, j = e - yb * P16; v[i] = src[(size_t)yb * G * P16 + j];
        }
    };
    int k = k0;
    if (k >= per) return;
    load(band * per + k);
    for (;;) {
        const int task = band * per + k;
#pragma unroll
        for (int i = 0; i < 4; i++) { const int e = threadIdx.x + i * 768; if (e < NV) lds[e] = v[i]; }
        __syncthreads();
        const int kn = k + nwg;
        if (kn < per && band * per + kn < ntask) load(band * per + kn);
        const int g = task % G, sub = task / G;
        float4* dst = out + (size_t)sub * unit + (size_t)g * NV;
        for (int e = threadIdx.x; e < NV; e += 768) dst[e] = lds[(e * 7) % NV];
        __syncthreads();
        k = kn;
        if (k >= per || band * per + k >= ntask) break;
    }
}

// contiguous 44.8 KB per workgroup, three structures: (0) load all -> LDS -> barrier -> store all (the kernels' phases),
// (1) registers straight to the stores, no LDS, no barrier, (2) as 0 with 256-thread workgroups taking 1/3 of the share each
template <int MODE, int T>
__global__ __launch_bounds__(T) void k_chunk_wg(const float4* __restrict__ in, float4* __restrict__ out, int NV, int ntask) {
    extern __shared__ float4 lds[];
    const int task = blockIdx.x;
    if (task >= ntask) return;
    const float4* src = in + (size_t)task * NV;
    float4* dst = out + (size_t)task * NV;
    constexpr int NI = (2800 + T - 1) / T;
    float4 v[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) { const int e = min((int)threadIdx.x + i * T, NV - 1); v[i] = src[e]; }
    if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < NI; i++) { const int e = threadIdx.x + i * T; if (e < NV) dst[e] = v[i]; }
        return;
    }
#pragma unroll
    for (int i = 0; i < NI; i++) { const int e = threadIdx.x + i * T; if (e < NV) lds[e] = v[i]; }
    __syncthreads();
    for (int e = threadIdx.x; e < NV; e += T) dst[e] = lds[NV - 1 - e];
}

__global__ __launch_bounds__(768) void k_pieces_pad(const float4* __restrict__ in, float4* __restrict__ out, int nb, int G, int GP, int nsub) {
    extern __shared__ float4 lds[];
    const int task = (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8);
    if (task >= G * nsub) return;
    const int g = task % G, sub = task / G;
    const float4* src = in + (size_t)sub * nb * GP * 8 + (size_t)g * 8;
    const int NV = nb * 8;
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int e = threadIdx.x + i * 768;
        if (e < NV) { const int yb = e >> 3, j = e & 7; v[i] = src[(size_t)yb * GP * 8 + j]; }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { const int e = threadIdx.x + i * 768; if (e < NV) lds[e] = v[i]; }
    __syncthreads();
    float4* dst = out + ((size_t)sub * G + g) * NV;
    for (int e = threadIdx.x; e < NV; e += 768) dst[e] = lds[NV - 1 - e];
}

// read-only / write-only halves of the two patterns
template <int SCATTER>
__global__ __launch_bounds__(768) void k_read_only(const float4* __restrict__ in, float* __restrict__ out, int nb, int G, int nsub) {
    const int task = (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8);
    if (task >= G * nsub) return;
    const int g = task % G, sub = task / G, NV = nb * 8;
    const float4* src = SCATTER ? in + (size_t)sub * nb * G * 8 + (size_t)g * 8 : in + (size_t)task * NV;
    float acc = 0.f;
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int e = min((int)threadIdx.x + i * 768, NV - 1);
        v[i] = SCATTER ? src[(size_t)(e >> 3) * G * 8 + (e & 7)] : src[e];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(768) void k_write_only(float4* __restrict__ out, int NV, int ntask) {
    const int task = blockIdx.x;
    if (task >= ntask) return;
    float4* dst = out + (size_t)task * NV;
    for (int e = threadIdx.x; e < NV; e += 768) dst[e] = make_float4(1.f, 2.f, 3.f, (float)e);
}

// the mirror image: contiguous read, scattered write (128-byte pieces at 22.5 KB stride); WMODE 1: non-temporal stores
template <int WMODE>
__global__ __launch_bounds__(768) void k_scatter_write(const float4* __restrict__ in, float4* __restrict__ out, int nb, int G, int nsub) {
    extern __shared__ float4 lds[];
    const int task = (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8);
    if (task >= G * nsub) return;
    const int g = task % G, sub = task / G, NV = nb * 8;
    const float4* src = in + (size_t)task * NV;
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { const int e = min((int)threadIdx.x + i * 768, NV - 1); v[i] = src[e]; }
#pragma unroll
    for (int i = 0; i < 4; i++) { const int e = threadIdx.x + i * 768; if (e < NV) lds[e] = v[i]; }
    __syncthreads();
    float4* dst = out + (size_t)sub * nb * G * 8 + (size_t)g * 8;
    typedef float v4 __attribute__((ext_vector_type(4)));
    for (int e = threadIdx.x; e < NV; e += 768) {
        const float4 w = lds[NV - 1 - e];
        float4* q = dst + (size_t)(e >> 3) * G * 8 + (e & 7);
        if (WMODE == 1) __builtin_nontemporal_store(v4{w.x, w.y, w.z, w.w}, reinterpret_cast<v4*>(q)); else *q = w;
    }
}
// scattered read with non-temporal loads + contiguous non-temporal stores
template <int NTL, int NTS>
__global__ __launch_bounds__(768) void k_pieces_nt(const float4* __restrict__ in, float4* __restrict__ out, int nb, int G, int nsub) {
    extern __shared__ float4 lds[];
    const int task = (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8);
    if (task >= G * nsub) return;
    const int g = task % G, sub = task / G, NV = nb * 8;
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4* src = reinterpret_cast<const v4*>(in) + (size_t)sub * nb * G * 8 + (size_t)g * 8;
    v4 v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { const int e = min((int)threadIdx.x + i * 768, NV - 1); const v4* q = src + (size_t)(e >> 3) * G * 8 + (e & 7); v[i] = NTL ? __builtin_nontemporal_load(q) : *q; }
    v4* l4 = reinterpret_cast<v4*>(lds);
#pragma unroll
    for (int i = 0; i < 4; i++) { const int e = threadIdx.x + i * 768; if (e < NV) l4[e] = v[i]; }
    __syncthreads();
    v4* dst = reinterpret_cast<v4*>(out) + ((size_t)sub * G + g) * NV;
    for (int e = threadIdx.x; e < NV; e += 768) { if (NTS) __builtin_nontemporal_store(l4[NV - 1 - e], dst + e); else dst[e] = l4[NV - 1 - e]; }
}

int main() {
    const int nsub = 64, L = 1400, HP = 704;
    const size_t bytes = (size_t)nsub * L * HP * 8;                            // one half-spectrum array: 505 MB
    float4 *a, *b;
    float4* a2;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&a2, bytes * 2)); CK(hipMemset(a2, 1, bytes * 2));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // plain copy for reference
    {
        for (int rep = 0; rep < 3; rep++) { CK(hipEventRecord(e0)); CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice)); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("hipMemcpy D2D            : %.3f ms  %.2f TB/s (read + write)\n", ms, 2 * bytes / ms / 1e9);
    }
#define RUN(P16, XCD)                                                                                                      \
    {                                                                                                                      \
        const int G = 176;                      /* column groups per sub-image: a workgroup's share stays ~45 KB */       \
        const int nbb = (int)(bytes / nsub / ((size_t)G * P16 * 16));                                                      \
        const double moved = 2.0 * nsub * (double)nbb * G * P16 * 16;                                                      \
        CK(hipFuncSetAttribute((const void*)k_pieces<P16>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));            \
        float best = 1e9;                                                                                                  \
        for (int rep = 0; rep < 5; rep++) {                                                                                \
            CK(hipEventRecord(e0));                                                                                        \
            hipLaunchKernelGGL(k_pieces<P16>, dim3((G * nsub + 7) / 8 * 8), dim3(768), 65536, 0, a, b, nbb, G, nsub, XCD); \
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                                           \
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;                                      \
        }                                                                                                                  \
        printf("pieces of %4d B, %5d per workgroup, xcd order %d: %.3f ms  %.2f TB/s (read + write)\n", P16 * 16, nbb, XCD, best, moved / best / 1e9); \
    }
    RUN(8, 0) RUN(8, 1) RUN(16, 0) RUN(16, 1) RUN(32, 1) RUN(64, 1)
#define RUNP(P16, LDSB, WGS)                                                                                               \
    {                                                                                                                      \
        const int G = 176;                                                                                                 \
        const int nbb = (int)(bytes / nsub / ((size_t)G * P16 * 16));                                                      \
        const double moved = 2.0 * nsub * (double)nbb * G * P16 * 16;                                                      \
        CK(hipFuncSetAttribute((const void*)k_pieces_persist<P16>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));     \
        float best = 1e9;                                                                                                  \
        for (int rep = 0; rep < 5; rep++) {                                                                                \
            CK(hipEventRecord(e0));                                                                                        \
            hipLaunchKernelGGL(k_pieces_persist<P16>, dim3(WGS), dim3(768), LDSB, 0, a, b, nbb, G, nsub);                  \
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                                           \
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;                                      \
        }                                                                                                                  \
        printf("persistent, next task prefetched, pieces of %4d B, %d B LDS, %d workgroups: %.3f ms  %.2f TB/s\n", P16 * 16, LDSB, WGS, best, moved / best / 1e9); \
    }
    RUNP(8, 65536, 512) RUNP(8, 49152, 768) RUNP(16, 65536, 512)
#define RUNC(MODE, T, LDSB, NVV, label)                                                                                    \
    {                                                                                                                      \
        const int ntask = (int)(bytes / 16 / NVV);                                                                         \
        const double moved = 2.0 * (double)ntask * NVV * 16;                                                               \
        CK(hipFuncSetAttribute((const void*)k_chunk_wg<MODE, T>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));       \
        float best = 1e9;                                                                                                  \
        for (int rep = 0; rep < 5; rep++) {                                                                                \
            CK(hipEventRecord(e0));                                                                                        \
            hipLaunchKernelGGL((k_chunk_wg<MODE, T>), dim3(ntask), dim3(T), LDSB, 0, a, b, NVV, ntask);                    \
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));                                                           \
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;                                      \
        }                                                                                                                  \
        printf("%s: %.3f ms  %.2f TB/s\n", label, best, moved / best / 1e9);                                               \
    }
    // the stride between a workgroup's pieces: [yb][GP][128 B] with GP >= 176 groups allocated per row block (padding)
    for (int GP : {176, 177, 178, 179, 180, 182, 184, 188, 192, 200, 208, 224, 256}) {
        const int G = 176, nbb = 350;
        const double moved = 2.0 * nsub * (double)nbb * G * 128;
        if ((size_t)nsub * nbb * GP * 128 > bytes * 2) continue;
        float best = 1e9;
        CK(hipFuncSetAttribute((const void*)k_pieces<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_pieces_pad, dim3((G * nsub + 7) / 8 * 8), dim3(768), 65536, 0, a2, b, nbb, G, GP, nsub);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("128-B pieces, %d groups allocated per row block (stride %d B): %.3f ms  %.2f TB/s\n", GP, GP * 128, best, moved / best / 1e9);
    }
    for (int wm = 0; wm < 5; wm++) {
        float best = 1e9;
        CK(hipFuncSetAttribute((const void*)k_scatter_write<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        CK(hipFuncSetAttribute((const void*)k_scatter_write<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        CK(hipFuncSetAttribute((const void*)k_pieces_nt<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        CK(hipFuncSetAttribute((const void*)k_pieces_nt<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        CK(hipFuncSetAttribute((const void*)k_pieces_nt<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0));
            if (wm == 0) hipLaunchKernelGGL(k_scatter_write<0>, dim3(176 * nsub), dim3(768), 65536, 0, a, b, 350, 176, nsub);
            else if (wm == 1) hipLaunchKernelGGL(k_scatter_write<1>, dim3(176 * nsub), dim3(768), 65536, 0, a, b, 350, 176, nsub);
            else if (wm == 2) hipLaunchKernelGGL((k_pieces_nt<1, 1>), dim3(176 * nsub), dim3(768), 65536, 0, a, b, 350, 176, nsub);
            else if (wm == 3) hipLaunchKernelGGL((k_pieces_nt<1, 0>), dim3(176 * nsub), dim3(768), 65536, 0, a, b, 350, 176, nsub);
            else hipLaunchKernelGGL((k_pieces_nt<0, 1>), dim3(176 * nsub), dim3(768), 65536, 0, a, b, 350, 176, nsub);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%s: %.3f ms  %.2f TB/s\n", wm == 0 ? "contiguous read, scattered 128-B writes" : wm == 1 ? "contiguous read, scattered 128-B non-temporal writes" : wm == 2 ? "scattered nt reads, contiguous nt writes" : wm == 3 ? "scattered nt reads, contiguous plain writes" : "scattered plain reads, contiguous nt writes", best, 2.0 * bytes / best / 1e9);
    }
    for (int sc = 0; sc < 2; sc++) {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0));
            if (sc) hipLaunchKernelGGL(k_read_only<1>, dim3(176 * nsub), dim3(768), 0, 0, a, (float*)b, 350, 176, nsub);
            else hipLaunchKernelGGL(k_read_only<0>, dim3(176 * nsub), dim3(768), 0, 0, a, (float*)b, 350, 176, nsub);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("read only, %s: %.3f ms  %.2f TB/s\n", sc ? "128-B pieces at 22.5 KB stride" : "contiguous 44.8 KB per workgroup", best, bytes / best / 1e9);
    }
    {
        float best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_write_only, dim3(176 * nsub), dim3(768), 0, 0, b, 2800, 176 * nsub);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("write only, contiguous 44.8 KB per workgroup: %.3f ms  %.2f TB/s\n", best, bytes / best / 1e9);
    }
    RUNC(0, 768, 65536, 2800, "contiguous 44.8 KB per workgroup of 768, load -> LDS -> barrier -> store, 2 per CU")
    RUNC(1, 768, 65536, 2800, "contiguous 44.8 KB per workgroup of 768, registers -> store (no LDS, no barrier), 2 per CU")
    RUNC(1, 768, 0, 2800, "contiguous 44.8 KB per workgroup of 768, registers -> store, LDS not limiting")
    RUNC(0, 256, 16384, 933, "contiguous 14.9 KB per workgroup of 256, load -> LDS -> barrier -> store, 8 per CU")
    RUNC(1, 256, 0, 933, "contiguous 14.9 KB per workgroup of 256, registers -> store")
    return 0;
}
