// does gfx950 execute DPP wave_shr:1 / wave_shl:1 across the whole 64-lane wave?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o) {
    const int v = threadIdx.x + 100;
    o[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);
    o[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);
}
int main() {
    int* d; hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 1; i < 64; i++) if (h[i] != 100 + i - 1) bad++;
    for (int i = 0; i < 63; i++) if (h[64 + i] != 100 + i + 1) bad++;
    printf("shr: %d %d %d %d %d %d | shl: %d %d %d %d %d  bad=%d\n", h[0], h[1], h[15], h[16], h[17], h[63], h[64], h[64 + 15], h[64 + 16], h[64 + 62], h[64 + 63], bad);
    return bad != 0;
}
