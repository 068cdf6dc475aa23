"""GPU box (scratch library tools/exp/_var/boxv, BBX_DBG_BOX_NOFALLBACK=1): which boxes of the bench frame the bracket kernel
leaves to the full sort, and why (1 few samples, 2 bracket overflow, 3 wings overflow, 4 median outside the bracket, 5 clip limit
inside the middle)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import ctypes as C
import numpy as np, torch
import bench
from blackbox_amd import reduce as R, _lib
ctx = R.Context(0)
raw, flat, bpm, ex = bench.synth_frame_device(torch, ctx.device, 5280, 1320, 20, 180, 4000, 'u16', extras=True, ntrans=50)
geom = R.geometry(raw.shape, 5280, 1320)
h, hm = {}, {}
R.gain_corr(h, 'ML1')
sol = R.os_solve(ctx, raw, h, 'ML1', geom)
data, mask = R.calibrate(ctx, raw, sol, h, hm, 'ML1', geom, mflat=flat, bpm=bpm)
R.mask_init_finish(ctx, mask, h, hm, geom)
ny, nx = data.shape
box = 60
m = torch.empty((ny // box, nx // box), dtype=torch.float32, device=ctx.device)
s = torch.empty_like(m)
_lib.check(_lib.lib.bbx_bkg_boxstats(ctx.h, ny, nx, box, C.c_void_p(data.data_ptr()), C.c_void_p(mask.data_ptr()), None, 0.5,
                                     C.c_void_p(m.data_ptr()), C.c_void_p(s.data_ptr()), ctx.stream()), 'boxstats')
ctx.sync()
sb = s.cpu().numpy().view(np.uint32)
mh = m.cpu().numpy()
fail = sb == 0x7fc0b0b0
print('boxes', fail.size, 'left to the full sort', int(fail.sum()))
for why in (1, 2, 3, 4, 5):
    print('  reason', why, int((fail & (mh == why)).sum()))
