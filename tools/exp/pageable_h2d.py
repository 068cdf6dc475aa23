"""GPU box: does a small hipMemcpyAsync from PAGEABLE host memory return before the stream's earlier work has finished?
(a busy stream, then the copy: host time of the call, for a pageable and a pinned source)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import ctypes as C
import numpy as np, torch
from blackbox_amd import reduce as R, _lib
ctx = R.Context(0)
a = torch.randn(8192, 8192, device=ctx.device)
dst = torch.empty(8192, dtype=torch.uint8, device=ctx.device)
src_page = np.arange(5760, dtype=np.int32).astype(np.uint8)
src_pin = torch.zeros(5760, dtype=torch.uint8).pin_memory()
st = torch.cuda.Stream(device=ctx.device)
for name, ptr in (('pageable', src_page.ctypes.data), ('pinned', src_pin.data_ptr()), ('pageable', src_page.ctypes.data), ('pinned', src_pin.data_ptr())):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        e0.record()
        for _ in range(10): b = a @ a                              # some ms of work ahead of the copy on the same stream
        e1.record()
        t0 = time.perf_counter()
        _lib.check(_lib.lib.bbx_copy_async(C.c_void_p(dst.data_ptr()), C.c_void_p(ptr), 5760, 0, ctx.stream()), 'copy')
        t1 = time.perf_counter()
    torch.cuda.synchronize()
    print('%-9s source: call took %8.1f us of host time; work queued ahead of it: %.2f ms' % (name, (t1 - t0) * 1e6, e0.elapsed_time(e1)))
