"""Full-size timings of every stage beyond bench.py's headline workload (BASELINE configs 3
and 5): crosstalk, satellite trails, mask counts, edge fill, background mesh, ZOGY with 64
sub-images of 1400^2, PSF photometry, master stack.  GPU box only.

    python tools/stage_bench.py [--reps 5]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

import numpy as np


def timed(torch, fn, reps):
    fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--skip-zogy', action='store_true')
    args = ap.parse_args()
    import torch
    import bench
    from blackbox_amd import masters, reduce as R, zogy as G
    ctx = R.Context(0)
    dev = ctx.device
    ysz, xsz, os_y, os_x = 5280, 1320, 20, 180
    raw, flat, bpm = bench.synth_frame_device(torch, dev, ysz, xsz, os_y, os_x, 2000, 'u16')
    geom = R.geometry(raw.shape, ysz, xsz)
    N = 2 * ysz * 8 * xsz
    out = {}
    header, hm = {}, {}
    R.gain_corr(header, 'ML1')
    sol = R.os_solve(ctx, raw, header, 'ML1', geom)
    data, mask = R.calibrate(ctx, raw, sol, header, hm, 'ML1', geom, mflat=flat, bpm=bpm)
    R.mask_init_finish(ctx, mask, header, hm, geom)
    R.cosmics_corr(ctx, data, header, mask, hm, 'ML1')
    ctx.sync()
    rs = np.random.RandomState(0)
    coeffs = np.zeros((16, 16)); coeffs[~np.eye(16, dtype=bool)] = rs.uniform(0, 2e-4, 240)
    GB = 1e-9
    t = timed(torch, lambda: R.xtalk_corr(ctx, data, coeffs, mask, geom), args.reps)
    out['xtalk'] = dict(ms=t, algorithmic_GB=9 * N * GB, GBps=9 * N * GB / (t * 1e-3))
    t = timed(torch, lambda: R.sat_detect(ctx, data, {}, mask, {}), args.reps)
    out['sat_trails'] = dict(ms=t)
    t = timed(torch, lambda: R.mask_header(ctx, mask, {}), args.reps)
    out['mask_header(incl. D2H sync)'] = dict(ms=t, algorithmic_GB=N * GB)
    t = timed(torch, lambda: R.edge_fill(ctx, data, mask, geom), args.reps)
    out['edge_fill'] = dict(ms=t)
    t = timed(torch, lambda: G.get_back(ctx, data, mask), args.reps)
    out['bkg_get_back(median+std mini)'] = dict(ms=t, algorithmic_GB=5 * N * GB)
    mini, mini_std = G.get_back(ctx, data, mask)
    work = data.clone()
    t = timed(torch, lambda: G.mini2back(ctx, mini, data.shape, subtract_from=work, want_bkg=False), args.reps)
    out['bkg_mini2back+subtract'] = dict(ms=t, algorithmic_GB=8 * N * GB)
    frames = [flat * (1 + 0.01 * k) for k in range(15)]
    t = timed(torch, lambda: masters.master_median(ctx, frames, 'flat', medsec=[1.0 + 0.01 * k for k in range(15)], bpm=bpm), 2)
    out['master_flat_15'] = dict(ms=t, algorithmic_GB=(15 * 4 + 1 + 4) * N * GB, GBps=(15 * 4 + 5) * N * GB / (t * 1e-3))
    del frames
    if not args.skip_zogy:
        ref = data * 0.9 + 3.0
        S = 25
        y, x = np.mgrid[0:S, 0:S] - S // 2
        p = (1 + (y * y + x * x) / 4.0) ** -2.5
        psf = torch.from_numpy(np.repeat((p / p.sum()).astype(np.float32)[None], 64, 0)).to(dev)
        t0 = time.perf_counter()
        res = G.optimal_subtraction(ctx, data, ref, mask, mask, psf, psf)
        ctx.sync()
        first = 1e3 * (time.perf_counter() - t0)
        t0 = time.perf_counter()
        res = G.optimal_subtraction(ctx, data, ref, mask, mask, psf, psf)
        ctx.sync()
        out['optimal_subtraction(2 frames bkg + 64x1400^2 ZOGY + transients)'] = dict(ms=1e3 * (time.perf_counter() - t0), first_call_ms=first,
                                                                                 ntrans=len(res['transients']))
        subs = [G.cut_subimages(ctx, a) for a in (data, ref)]
        L = subs[0].shape[1]
        P = G.embed_psfs(ctx, psf, L)
        V = torch.ones_like(subs[0]) * 300
        sc = np.tile(np.array([[18, 9, 1, 1, 0.03, 0.03]], np.float32), (64, 1))
        t = timed(torch, lambda: G.run_zogy(ctx, subs[0], subs[1], P, P, V, V, sc), 3)
        out['run_zogy_64x%d' % L] = dict(ms=t, io_model_GB=34 * N * GB)
    from blackbox_amd import fpack as P
    t0 = time.perf_counter(); cd = P.compress_tiles(ctx, data, 16, 1); ctx.sync(); t1 = time.perf_counter()
    cd = P.compress_tiles(ctx, data, 16, 1, _view=True); ctx.sync(); t2 = time.perf_counter()   # as fpack_image calls it: heap stays in the pinned buffer
    nheap, nlossless = cd['heap'].size, int((cd['flag'] != 0).sum())
    cm = P.compress_tiles(ctx, mask); ctx.sync(); t3 = time.perf_counter()
    hm = cm['heap']
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rnd = P._rnd(dev)
    stride = P.lib.bbx_fpack_tile_stride(data.shape[1], 4)
    scratch = torch.empty(data.shape[0] * stride, dtype=torch.uint8, device=dev)
    tiles = torch.empty(data.shape[0] * 24, dtype=torch.uint8, device=dev)
    import ctypes as C
    ev0.record()
    for _ in range(3):
        P.check(P.lib.bbx_fpack_tiles(ctx.h, data.shape[0], data.shape[1], C.c_void_p(data.data_ptr()), -32, 16.0, 1,
                                      C.c_void_p(rnd.data_ptr()), C.c_void_p(scratch.data_ptr()), C.c_void_p(tiles.data_ptr()),
                                      ctx.stream()), 'bbx_fpack_tiles', ctx.h)
    ev1.record(); torch.cuda.synchronize()
    out['fpack_q16_float_frame'] = dict(kernel_ms=ev0.elapsed_time(ev1) / 3, end_to_end_ms_incl_D2H=1e3 * (t2 - t1),
                                        compressed_MB=nheap / 1e6, ratio=data.numel() * 4 / nheap,
                                        rows_stored_losslessly=nlossless)
    out['fpack_mask'] = dict(end_to_end_ms_incl_D2H=1e3 * (t3 - t2), compressed_MB=hm.size / 1e6, ratio=mask.numel() / hm.size)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        pz = P.fpack_image(ctx, os.path.join(td, 'raw.fits'), raw)
        P.funpack_image(ctx, pz)
        t0 = time.perf_counter(); back, _ = P.funpack_image(ctx, pz); ctx.sync(); t1 = time.perf_counter()
        assert torch.equal(back, raw)
        out['funpack_raw_u16_frame'] = dict(end_to_end_ms_incl_file_read_H2D=1e3 * (t1 - t0), file_MB=os.path.getsize(pz) / 1e6,
                                            raw_MB=raw.numel() * 2 / 1e6)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
