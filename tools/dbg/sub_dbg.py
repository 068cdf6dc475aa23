"""bench scene: optimal_subtraction with the reference given as a co-add (bkg-subtracted + std mini) or raw (measured)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, torch
import bench
from blackbox_amd import reduce as R, zogy as G
from blackbox_amd._lib import lib

if __name__ == '__main__':
    ctx = R.Context(0)
    dev = ctx.device
    ysz, xsz, os_y, os_x = 5280, 1320, 20, 180
    raw, flat, bpm, ex = bench.synth_frame_device(torch, dev, ysz, xsz, os_y, os_x, 4000, 'u16', extras=True, ntrans=50)
    ref, ref_mask = bench.synth_reference(torch, dev, ex.pop('scene0'), 4000)
    rs = np.random.RandomState(0)
    coeffs = np.zeros((16, 16)); coeffs[~np.eye(16, dtype=bool)] = rs.uniform(0, 2e-4, 240)
    data, mask, header, hm = R.reduce_object(ctx, raw, {}, 'ML1', mflat=flat, bpm=bpm, xtalk_coeffs=coeffs, exptime=60.0)
    p = torch.from_numpy(bench.moffat_stamp(25, 4.0)).to(dev)
    std8 = np.full((176, 176), 8.0, np.float32)
    keep = {}
    for core in (1,):
        lib.bbx_set_option(ctx.h, 3, core)
        for name, kw in (('measured', dict(ref_is_bkgsub=False)), ('bkgsub+mini', dict(ref_is_bkgsub=True, ref_bkg_std_mini=std8)),
                         ('bkgsub', dict(ref_is_bkgsub=True))):
            res = G.optimal_subtraction(ctx, data, ref, mask, ref_mask, p, p, fratio=1.0, dx=0.03, dy=0.03, cat_extract=False, **kw)
            ctx.sync()
            h = res['header_trans']
            print(core, name, 'SCSTD', h['Z-SCSTD'][0], 'SCMED', h['Z-SCMED'][0], 'ntrans', h['T-NTRANS'][0], 'S-BKGSTDR', h['S-BKGSTDR'][0],
                  'Dstd', float(res['D'][2000:2200, 2000:2200].std()), 'Fpsferr', float(res['Fpsferr'][2100, 2100]), 'rptr%16', res['ref_bkgsub'].data_ptr() % 16, res['ref_bkgsub'].dtype, res['ref_bkgsub'].is_contiguous(), 'rbstd', float(res['bkg_std_ref'][2100,2100]),
                  'sr', res['scal'][:3, 1], 'sn', res['scal'][:3, 0], 'ref std', float(res['ref_bkgsub'][2000:2200, 2000:2200].std()), flush=True)
            keep[name] = res
    a, b = keep['measured'], keep['bkgsub']
    for k in ('ref_bkgsub', 'bkg_std_ref', 'data_bkgsub', 'bkg_std', 'D', 'Scorr', 'Fpsferr'):
        d = (a[k] - b[k]).abs()
        print(k, 'max diff', float(d.max()), 'mean', float(d.mean()), flush=True)
    print('ref vs ref_bkgsub(bkgsub) same object', b['ref_bkgsub'].data_ptr() == ref.data_ptr())
