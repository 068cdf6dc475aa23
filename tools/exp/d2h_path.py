"""GPU box: which path a 100 MB pinned D2H copy takes from Python (run under rocprofv3 --kernel-trace --stats: a blit shows as
__amd_rocclr_copyBuffer).  argv[1]: torch | hip | bbx | hipmalloc"""
import ctypes as C, os, sys, time
import torch
mode = sys.argv[1]
n = 100 << 20
d = torch.empty(n, dtype=torch.uint8, device='cuda')
h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
s = torch.cuda.Stream()
sp = C.c_void_p(s.cuda_stream)
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
dptr = d.data_ptr()
if mode == 'hipmalloc':                       # a device buffer of its own (not a piece of torch's caching allocator's block)
    p = C.c_void_p()
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    assert hip.hipMalloc(C.byref(p), n) == 0
    dptr = p.value
hptr = h.data_ptr()
if mode == 'hiphost':                         # destination: hipHostMalloc'ed here, not torch's pinned allocator
    p = C.c_void_p()
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    assert hip.hipHostMalloc(C.byref(p), n, 0) == 0
    hptr = p.value
def copy():
    if mode == 'torch':
        with torch.cuda.stream(s):
            h.copy_(d, non_blocking=True)
    elif mode in ('hip', 'hipmalloc', 'hiphost'):
        assert hip.hipMemcpyAsync(hptr, dptr, n, 2, sp) == 0          # hipMemcpyDeviceToHost = 2
    else:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
        from blackbox_amd._lib import lib
        assert lib.bbx_copy_async(C.c_void_p(h.data_ptr()), C.c_void_p(dptr), n, 1, sp) == 0
for rep in range(3):
    copy()
torch.cuda.synchronize()
t0 = time.perf_counter()
for rep in range(5):
    copy()
torch.cuda.synchronize()
import os as _os
print({k: v for k, v in _os.environ.items() if k.startswith(('HSA_', 'HIP_', 'GPU_', 'ROC', 'AMD_', 'PYTORCH'))})
print('%s: D2H 100 MB x5: %.1f GB/s' % (mode, 5 * n / 1e9 / (time.perf_counter() - t0)))
