"""GPU box: where the host time of a frame goes (the Python side of reduce_object + optimal_subtraction, serial frames on one
stream): cProfile sorted by own time, and CPU seconds per frame of this thread"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np


def main():
    import torch
    import bench
    from blackbox_amd import reduce as R, zogy as G
    ctx = R.Context(0)
    dev = ctx.device
    ysz, xsz = 5280, 1320
    raw, flat, bpm, ex = bench.synth_frame_device(torch, dev, ysz, xsz, 20, 180, 4000, 'u16', extras=True, ntrans=50)
    ref, ref_mask = bench.synth_reference(torch, dev, ex.pop('scene0'), 4000)
    rs = np.random.RandomState(0)
    coeffs = np.zeros((16, 16)); coeffs[~np.eye(16, dtype=bool)] = rs.uniform(0, 2e-4, 240)
    zi = bench.zogy_inputs(torch, dev, 8, 8, 49, 60, 2 * ysz, 8 * xsz)

    def frame():
        data, mask, h, hm = R.reduce_object(ctx, raw, {}, 'ML1', mflat=flat, bpm=bpm, xtalk_coeffs=coeffs, exptime=60.0)
        res = G.optimal_subtraction(ctx, data, ref, mask, ref_mask, cat_extract=True, **zi)
        ctx.sync()
    for _ in range(3):
        frame()
    n = 20
    c0, w0 = time.thread_time(), time.perf_counter()
    for _ in range(n):
        frame()
    print('per frame: wall %.2f ms, CPU of this thread %.2f ms' % ((time.perf_counter() - w0) / n * 1e3, (time.thread_time() - c0) / n * 1e3))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(n):
        frame()
    pr.disable()
    st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(45)


if __name__ == '__main__':
    main()
