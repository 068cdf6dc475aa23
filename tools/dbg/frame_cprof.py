"""GPU box: wall-clock profile (cProfile, cumulative) of one serial frame of the headline workload -- where the time of
`blackbox.py --image F`'s compute goes on the host side (waits on the device show up in the .cpu() / sync calls)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import torch
import bench
from blackbox_amd import reduce as R, zogy as G

ctx = R.Context(0)
dev = ctx.device
raw, flat, bpm, ex = bench.synth_frame_device(torch, dev, 5280, 1320, 20, 180, 4000, 'u16', extras=True, ntrans=50)
ref, ref_mask = bench.synth_reference(torch, dev, ex.pop('scene0'), 4000)
rs = np.random.RandomState(0)
coeffs = np.zeros((16, 16)); coeffs[~np.eye(16, dtype=bool)] = rs.uniform(0, 2e-4, 240)
psf = torch.from_numpy(bench.moffat_stamp(25, 4.0)).to(dev)


def frame():
    data, mask, h, hm = R.reduce_object(ctx, raw, {}, 'ML1', mflat=flat, bpm=bpm, xtalk_coeffs=coeffs, exptime=60.0)
    res = G.optimal_subtraction(ctx, data, ref, mask, ref_mask, psf, psf, fratio=1.0, dx=0.03, dy=0.03, cat_extract=True,
                                ref_is_bkgsub=True, ref_bkg_std_mini=np.full((176, 176), 8.0, np.float32))
    ctx.sync()
    return res


for i in range(3):
    frame()
ts = []
for i in range(8):
    t0 = time.perf_counter(); frame(); ts.append(time.perf_counter() - t0)
print('serial frame ms: min %.2f median %.2f' % (1e3 * min(ts), 1e3 * sorted(ts)[len(ts) // 2]))
pr = cProfile.Profile()
pr.enable()
for i in range(5):
    frame()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
st.sort_stats('tottime').print_stats(25)
