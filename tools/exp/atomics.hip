// Throughput of returning global atomics: one address vs several addresses (spacing sweep).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(unsigned* ctr, int naddr, int stride_words, int per_block, unsigned* sink) {
    // one returning atomic per block per iteration (thread 0), like a list reservation
    unsigned acc = 0;
    if (threadIdx.x == 0) {
        for (int i = 0; i < per_block; i++) {
            const int a = (blockIdx.x + i) % naddr;
            acc += atomicAdd(&ctr[(size_t)a * stride_words], 1u);
        }
        if (acc == 0xffffffffu) sink[0] = acc;
    }
}
__global__ __launch_bounds__(256) void k_nr(unsigned* ctr, int naddr, int stride_words, int per_block) {
    if (threadIdx.x == 0)
        for (int i = 0; i < per_block; i++) atomicAdd(&ctr[(size_t)((blockIdx.x + i) % naddr) * stride_words], 1u);   // non-returning
}
int main() {
    unsigned *c, *sink; hipMalloc(&c, 64 << 20); hipMalloc(&sink, 64); hipMemset(c, 0, 64 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 4096, per = 8;
    for (int ret = 1; ret >= 0; ret--)
    for (int stride : {1, 16, 64, 1024, 16384})
        for (int naddr : {1, 2, 8, 32, 128}) {
            if (naddr == 1 && stride != 1) continue;
            auto go = [&] { if (ret) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, c, naddr, stride, per, sink);
                            else hipLaunchKernelGGL(k_nr, dim3(blocks), dim3(256), 0, 0, c, naddr, stride, per); };
            go(); hipDeviceSynchronize();
            hipEventRecord(e0); for (int r = 0; r < 5; r++) go(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            printf("%s naddr=%3d stride=%6d B : %8.1f us  %6.2f ns/atomic\n", ret ? "ret  " : "noret", naddr, stride * 4, ms * 1e3, ms * 1e6 / (blocks * per));
        }
    return 0;
}
