"""GPU box: bbx_bkg_boxstats on a full-size synthetic frame, bracket path against the full sort (BBX_OPT_BKG_FULL_SORT): time,
equality of the medians, largest relative difference of the std"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import ctypes as C
import numpy as np, torch
import bench
from blackbox_amd import reduce as R, _lib
ctx = R.Context(0)
raw, flat, bpm = bench.synth_frame_device(torch, ctx.device, 5280, 1320, 20, 180, 2000, 'u16')
geom = R.geometry(raw.shape, 5280, 1320)
h, hm = {}, {}
R.gain_corr(h, 'ML1')
sol = R.os_solve(ctx, raw, h, 'ML1', geom)
data, mask = R.calibrate(ctx, raw, sol, h, hm, 'ML1', geom, mflat=flat, bpm=bpm)
R.mask_init_finish(ctx, mask, h, hm, geom)
ny, nx = data.shape
box = 60
res = {}
for full in (1, 0, 1, 0):
    _lib.check(_lib.lib.bbx_set_option(ctx.h, 8, full), 'opt')
    m = torch.empty((ny // box, nx // box), dtype=torch.float32, device=ctx.device)
    s = torch.empty_like(m)
    def run():
        _lib.check(_lib.lib.bbx_bkg_boxstats(ctx.h, ny, nx, box, C.c_void_p(data.data_ptr()), C.c_void_p(mask.data_ptr()), None, 0.5,
                                             C.c_void_p(m.data_ptr()), C.c_void_p(s.data_ptr()), ctx.stream()), 'boxstats')
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    res[full] = (m.cpu().numpy(), s.cpu().numpy())
    print('full_sort=%d  %.3f ms per frame' % (full, e0.elapsed_time(e1) / 20))
a, b = res[0], res[1]
ok = ~np.isnan(b[0])
print('medians equal:', np.array_equal(a[0][ok], b[0][ok]) and np.array_equal(np.isnan(a[0]), np.isnan(b[0])),
      ' std max rel diff: %.2e' % np.max(np.abs(a[1][ok] - b[1][ok]) / b[1][ok]))
