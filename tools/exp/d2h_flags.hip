// GPU box: does a 100 MB device-to-host copy run on a copy engine or as a blit kernel, by kind of host memory?
// (run under rocprofv3 --kernel-trace --stats: blits show as __amd_rocclr_copyBuffer)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sys/mman.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const size_t n = 100u << 20;
    void *d, *h = nullptr;
    CK(hipMalloc(&d, n));
    CK(hipMemset(d, 1, n));
    const char* what = "";
    if (mode == 0) { CK(hipHostMalloc(&h, n, hipHostMallocDefault)); what = "hipHostMallocDefault"; }
    if (mode == 1) { CK(hipHostMalloc(&h, n, hipHostMallocNonCoherent)); what = "hipHostMallocNonCoherent"; }
    if (mode == 2) { CK(hipHostMalloc(&h, n, hipHostMallocCoherent)); what = "hipHostMallocCoherent"; }
    if (mode == 3) { h = aligned_alloc(4096, n); CK(hipHostRegister(h, n, hipHostRegisterDefault)); what = "hipHostRegister(malloc)"; }
    if (mode == 4) { CK(hipHostMalloc(&h, n, hipHostMallocNumaUser)); what = "hipHostMallocNumaUser"; }
    if (mode == 5) { CK(hipHostMalloc(&h, n, hipHostMallocPortable | hipHostMallocMapped)); what = "Portable|Mapped"; }
    hipStream_t s;
    if (argc > 2) { CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, 0)); what = "default host memory, non-blocking priority stream"; }
    else CK(hipStreamCreate(&s));
    for (int r = 0; r < 2; r++) { CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); }
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 5; r++) CK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("mode %d %-28s D2H %.1f GB/s\n", mode, what, 5.0 * n / dt / 1e9);
    return 0;
}
