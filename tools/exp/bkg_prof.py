import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
from blackbox_amd import reduce as R, zogy as G
ctx = R.Context(0)
raw, flat, bpm = bench.synth_frame_device(torch, ctx.device, 5280, 1320, 20, 180, 2000, 'u16')
geom = R.geometry(raw.shape, 5280, 1320)
h, hm = {}, {}
R.gain_corr(h, 'ML1')
sol = R.os_solve(ctx, raw, h, 'ML1', geom)
data, mask = R.calibrate(ctx, raw, sol, h, hm, 'ML1', geom, mflat=flat, bpm=bpm)
R.mask_init_finish(ctx, mask, h, hm, geom)
for _ in range(4):
    mini, mstd = G.get_back(ctx, data, mask)
work = data.clone()
for _ in range(4):
    G.mini2back(ctx, mini, data.shape, subtract_from=work, want_bkg=False)
ctx.sync()
