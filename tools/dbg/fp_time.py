"""GPU box: time bbx_fpack_tiles on full-size frames: float (q = 16 / 4 / 2) with the bracket / the histogram medians
(BBX_OPT_FPACK_HIST_ONLY) and the short / worst-case stream buffer (BBX_OPT_FPACK_ONE_WG), and the uint8 mask; checks that
all variants make the same bytes"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import ctypes as C
import numpy as np, torch
from blackbox_amd import reduce as R, fpack as P, _lib
ctx = R.Context(0)
g = torch.Generator(device=ctx.device); g.manual_seed(1)
img = (300 + 9 * torch.randn(10560, 10560, device=ctx.device, generator=g)).contiguous()
img[:30] = 251.0; img[-30:] = 251.0                              # constant edge rows (refused by the quantiser)
img[4000:4003, 100:9000] += 5e4                                   # bright rows
msk = torch.zeros((10560, 10560), dtype=torch.uint8, device=ctx.device)
msk[torch.rand(10560, 10560, device=ctx.device, generator=g) < 0.002] = 2
msk[:30] = 32; msk[-30:] = 32; msk[:, :30] = 32; msk[:, -30:] = 32
ny, nx = img.shape
stride = _lib.lib.bbx_fpack_tile_stride(nx, 4)
scratch = torch.empty(ny * stride, dtype=torch.uint8, device=ctx.device)
tiles = torch.empty(ny * 24, dtype=torch.uint8, device=ctx.device)
rnd = P._rnd(ctx.device)


def t(f, n=10):
    for _ in range(2): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def run(a, bitpix, q):
    _lib.check(_lib.lib.bbx_fpack_tiles(ctx.h, ny, nx, C.c_void_p(a.data_ptr()), bitpix, float(q), 1, C.c_void_p(rnd.data_ptr()),
                                        C.c_void_p(scratch.data_ptr()), C.c_void_p(tiles.data_ptr()), ctx.stream()), 'tiles')


ref = {}
COMBOS = ((0, 0, 0),) if os.environ.get('FP_ONLY') else ((1, 1, 1), (1, 0, 1), (0, 1, 1), (0, 0, 0))       # FP_ONLY=1: the product configuration alone
for hist, one, tile in COMBOS:
    if True:
        _lib.check(_lib.lib.bbx_set_option(ctx.h, 6, hist), 'opt')
        _lib.check(_lib.lib.bbx_set_option(ctx.h, 5, one), 'opt')
        for name, a, bp, q in (('float q16', img, -32, 16), ('float q4', img, -32, 4), ('float q2', img, -32, 2), ('mask', msk, 8, 0)):
            ms = t(lambda: run(a, bp, q))
            torch.cuda.synchronize()
            tl = tiles.cpu().numpy().view(P._TILE_DT).copy()
            sc = scratch.view(ny, stride)
            # checksum of the streams: bytes beyond nbytes are stale -> mask them
            nb = torch.from_numpy(tl['nbytes'].astype(np.int64)).to(ctx.device)
            valid = torch.arange(stride, device=ctx.device)[None, :] < nb[:, None]
            h = int((sc.to(torch.int64) * valid * (1 + torch.arange(stride, device=ctx.device)[None, :] % 251)).sum().item())
            key = (name,)
            sig = (h, tl['nbytes'].sum(), tl['flag'].sum(), tl['zscale'].tobytes(), tl['zzero'].tobytes())
            same = ''
            if key in ref:
                same = 'same bytes' if ref[key] == sig else '*** DIFFERENT ***'
            else:
                ref[key] = sig
            print('hist_only=%d one_wg=%d tile_only=%d %-10s %.3f ms  %5.1f MB  refused rows %d  %s' % (hist, one, tile, name, ms, tl['nbytes'].sum() / 1e6, (tl['flag'] != 0).sum(), same))
_lib.check(_lib.lib.bbx_set_option(ctx.h, 6, 0), 'opt')
_lib.check(_lib.lib.bbx_set_option(ctx.h, 5, 0), 'opt')
