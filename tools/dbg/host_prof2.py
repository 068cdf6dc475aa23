"""cProfile of the host side of a frame (reduce_object + optimal_subtraction) on a small geometry, where the GPU work is negligible"""
import cProfile, pstats, os, sys, io
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
import bench
from blackbox_amd import reduce as R, zogy as G
ctx = R.Context(0)
dev = ctx.device
ysz, xsz = 660, 330
raw, flat, bpm, ex = bench.synth_frame_device(torch, dev, ysz, xsz, 20, 45, 4000, 'u16', extras=True, ntrans=50)
ref, ref_mask = bench.synth_reference(torch, dev, ex.pop('scene0'), 4000)
rs = np.random.RandomState(0)
coeffs = np.zeros((16, 16)); coeffs[~np.eye(16, dtype=bool)] = rs.uniform(0, 2e-4, 240)
psf = torch.from_numpy(bench.moffat_stamp(25, 4.0)).to(dev)
kw = dict(ref=ref, ref_mask=ref_mask, psf_new=psf, psf_ref=psf, fratio=1.0, dx=0.03, dy=0.03, ref_is_bkgsub=True,
          ref_bkg_std_mini=np.full((2 * ysz // 30, 8 * xsz // 30), 8.0, np.float32), cat_extract=True, trans_extract=True,
          subimage_size=330, subimage_border=20, bkg_boxsize=30)
def frame():
    data, mask, h, hm = R.reduce_object(ctx, raw, {}, 'ML1', mflat=flat, bpm=bpm, xtalk_coeffs=coeffs, exptime=60.0, ysize_chan=ysz, xsize_chan=xsz)
    return G.optimal_subtraction(ctx, data, new_mask=mask, **kw)
for _ in range(3): frame()
pr = cProfile.Profile(); pr.enable()
N = 20
for _ in range(N): frame()
pr.disable()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats('tottime'); ps.print_stats(28)
print('\n'.join(l for l in s.getvalue().splitlines() if l.strip())[:6000])
print('per frame: divide by', N)
