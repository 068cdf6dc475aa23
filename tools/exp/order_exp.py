"""GPU box: does the dense LA-Cosmic pass run faster right after k_calibrate (data still in the
memory-side cache) than after the mask stage?  Timing only (the early order uses an unfinished mask)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import ctypes as C
import torch
import bench
from blackbox_amd import reduce as R, _lib

ctx = R.Context(0)
ysz, xsz = 5280, 1320
raw, flat, bpm = bench.synth_frame_device(torch, ctx.device, ysz, xsz, 20, 180, 2000, 'u16')
geom = R.geometry(raw.shape, ysz, xsz)


def frame(early):
    header, hm = {}, {}
    R.gain_corr(header, 'ML1')
    sol = R.os_solve(ctx, raw, header, 'ML1', geom)
    data, mask = R.calibrate(ctx, raw, sol, header, hm, 'ML1', geom, mflat=flat, bpm=bpm)
    if early:
        R.cosmics_corr(ctx, data, header, mask, hm, 'ML1')
        R.mask_init_finish(ctx, mask, header, hm, geom)
    else:
        R.mask_init_finish(ctx, mask, header, hm, geom)
        R.cosmics_corr(ctx, data, header, mask, hm, 'ML1')
    ctx.sync()


for early in (False, True, False, True):
    frame(early)
    _lib.check(_lib.lib.bbx_profile_enable(ctx.h, 1), 'p')
    for _ in range(10):
        frame(early)
    ms = (C.c_double * 8)(); n = (C.c_int32 * 8)()
    _lib.check(_lib.lib.bbx_profile_read(ctx.h, ms, n, 8), 'r', ctx.h)
    _lib.check(_lib.lib.bbx_profile_enable(ctx.h, 0), 'p')
    print('dense pass right after calibrate' if early else 'dense pass after the mask stage ', 'k_lac_cand us', round(1e3 * ms[1] / n[1], 1),
          'k_calibrate us', round(1e3 * ms[0] / n[0], 1))
