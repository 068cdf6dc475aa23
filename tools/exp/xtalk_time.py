import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
import bench
from blackbox_amd import reduce as R
ctx = R.Context(0)
raw, flat, bpm = bench.synth_frame_device(torch, ctx.device, 5280, 1320, 20, 180, 2000, 'u16')
geom = R.geometry(raw.shape, 5280, 1320)
h, hm = {}, {}
R.gain_corr(h, 'ML1')
sol = R.os_solve(ctx, raw, h, 'ML1', geom)
data, mask = R.calibrate(ctx, raw, sol, h, hm, 'ML1', geom, mflat=flat, bpm=bpm)
rs = np.random.RandomState(0)
coeffs = np.zeros((16, 16)); coeffs[~np.eye(16, dtype=bool)] = rs.uniform(0, 2e-4, 240)
for vw, grid in (('1', 0), ('2', 0), ('4', 0)):
    os.environ['BBX_XTALK_VEC'] = vw
    os.environ['BBX_XTALK_GRID'] = str(grid)
    R.xtalk_corr(ctx, data, coeffs, mask, geom); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): R.xtalk_corr(ctx, data, coeffs, mask, geom)
    e1.record(); torch.cuda.synchronize()
    print('xtalk VEC', vw, 'grid', grid, round(e0.elapsed_time(e1) / 10, 3), 'ms')
