"""GPU box: which path a 100 MB pinned D2H copy takes (run under rocprofv3 --kernel-trace: a blit shows as __amd_rocclr_copyBuffer)"""
import time, torch
d = torch.empty(100 << 20, dtype=torch.uint8, device='cuda')
h = torch.empty(100 << 20, dtype=torch.uint8, pin_memory=True)
s = torch.cuda.Stream()
x = torch.randn(4096, 4096, device='cuda')
for rep in range(3):
    with torch.cuda.stream(s):
        y = x @ x                                   # compute in front of the copy on the same stream
        h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
t0 = time.perf_counter()
for rep in range(5):
    with torch.cuda.stream(s):
        h.copy_(d, non_blocking=True)
torch.cuda.synchronize()
print('D2H 100 MB x5: %.1f GB/s' % (5 * 100 * 1.048576 / 1e3 / (time.perf_counter() - t0)))
