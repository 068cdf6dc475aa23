"""GPU box: time bbx_fpack_body on a full-size float frame (q = 16) with the short / the worst-case stream buffer"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
from blackbox_amd import reduce as R, fpack as P, _lib
ctx = R.Context(0)
img = (300 + 9 * torch.randn(10560, 10560, device=ctx.device)).contiguous()
def t(f, n=6):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for one in (1, 0):
    _lib.check(_lib.lib.bbx_set_option(ctx.h, 5, one), 'opt')
    print('BBX_OPT_FPACK_ONE_WG=%d: compress_tiles (k_fp_tile + gather + copies) %.3f ms' % (one, t(lambda: P.compress_tiles(ctx, img, 16, 1, _view=True))))
import ctypes as C
ny, nx = img.shape
stride = _lib.lib.bbx_fpack_tile_stride(nx, 4)
scratch = torch.empty(ny * stride, dtype=torch.uint8, device=ctx.device)
tiles = torch.empty(ny * 24, dtype=torch.uint8, device=ctx.device)
rnd = P._rnd(ctx.device)
def tiles_only():
    _lib.check(_lib.lib.bbx_fpack_tiles(ctx.h, ny, nx, C.c_void_p(img.data_ptr()), -32, 16.0, 1, C.c_void_p(rnd.data_ptr()), C.c_void_p(scratch.data_ptr()),
                                        C.c_void_p(tiles.data_ptr()), ctx.stream()), 'tiles')
for one in (1, 0):
    _lib.check(_lib.lib.bbx_set_option(ctx.h, 5, one), 'opt')
    print('BBX_OPT_FPACK_ONE_WG=%d: bbx_fpack_tiles alone %.3f ms' % (one, t(tiles_only, 10)))
