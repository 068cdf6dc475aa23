"""GPU box: cProfile of serial frames of the headline workload (host-side time per frame)"""
import cProfile, pstats, os, sys, io
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
    psf = torch.from_numpy(bench.moffat_stamp(25, 4.0)).to(dev)
    std8 = np.full((176, 176), 8.0, np.float32)

    def frame():
        data, mask, h, hm = R.reduce_object(ctx, raw, {}, 'ML1', mflat=flat, bpm=bpm, xtalk_coeffs=coeffs, exptime=60.0)
        res = G.optimal_subtraction(ctx, data, ref, mask, ref_mask, psf, psf, fratio=1.0, dx=0.03, dy=0.03, cat_extract=True,
                                    ref_is_bkgsub=True, ref_bkg_std_mini=std8)
        ctx.sync()
    for i in range(2):
        frame()
    n = 5
    pr = cProfile.Profile()
    pr.enable()
    for i in range(n):
        frame()
    pr.disable()
    st = io.StringIO()
    pstats.Stats(pr, stream=st).sort_stats('tottime').print_stats(35)
    print(st.getvalue())
    print('frames', n)


if __name__ == '__main__':
    main()
