"""which step of the serial reduction raises a device error flag on the bench scene"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
import bench
from blackbox_amd import reduce as R
ctx = R.Context(0)
raw, flat, bpm, ex = bench.synth_frame_device(torch, ctx.device, 5280, 1320, 20, 180, 4000, 'u16', extras=True, ntrans=50)
rs = np.random.RandomState(0)
coeffs = np.zeros((16, 16)); coeffs[~np.eye(16, dtype=bool)] = rs.uniform(0, 2e-4, 240)
for rep in range(2):
    h = {}
    data, mask, h, hm = R.reduce_object(ctx, raw, h, 'ML1', mflat=flat, bpm=bpm, xtalk_coeffs=coeffs, exptime=60.0)
    print({k: R.hval(h, k) for k in ('OS-P', 'MASK-P', 'COSMIC-P', 'XTALK-P', 'SAT-P', 'NSATS', 'NCOSMICS', 'RDNOISE', 'BIASMEAN', 'NOBJ-SAT')})
    d_n, d_info = R.sat_detect(ctx, data, {}, mask.clone(), {})
    try:
        ctx.sync()
    except Exception as e:
        print('sync after sat_detect:', e)
    print('sat info', d_info.cpu().numpy())
