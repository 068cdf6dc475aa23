import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'oracle'))
import numpy as np, torch
import bench
from blackbox_amd import reduce as R, zogy as G, synth
import bbx_oracle as O
ctx = R.Context(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)
YS, XS = 120, 330
for seed in (77, 78, 79):
    case = synth.make_case(YS, XS, seed, tel='ML1', os_y=20, os_x=45, n_stars=60, n_sat=2, n_cr=40)
    d0, m0, h0, _ = R.reduce_object(ctx, dev(case['raw']), {}, 'ML1', mflat=dev(case['flat']), bpm=dev(case['bpm']),
                                    xtalk_coeffs=O.xtalk_coeffs(case['xtalk']), exptime=60.0, ysize_chan=YS, xsize_chan=XS)
    mini, mini_std = G.get_back(ctx, d0, m0, bkg_boxsize=30)
    work = d0.clone()
    G.mini2back(ctx, mini, d0.shape, bkg_boxsize=30, interp_Xchan=True, subtract_from=work, want_bkg=False)
    thr = 5 * float(np.median(mini_std.cpu().numpy()))
    print(seed, 'thr', thr, 'n above', int((work.abs() >= thr).sum().item()), 'npix', work.numel(), 'cap', work.numel() // 16 + 1024,
          'nan', int(torch.isnan(work).sum().item()), 'bkg', float(np.median(mini.cpu().numpy())))
    try:
        p = G.find_transients(ctx, work, thr, max_out=200000)
        print('  peaks', len(p))
    except Exception as e:
        print('  find failed', e)
