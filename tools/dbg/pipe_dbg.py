import os, sys, logging
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'oracle'))
import numpy as np, torch
import bench
from blackbox_amd import reduce as R, zogy as G, synth, _lib
from blackbox_amd.pipeline import FramePipeline, HostPool
import bbx_oracle as O
def main():
    logging.basicConfig(level='INFO')
    pool = HostPool(2)
    ctx = R.Context(0)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(ctx.device)
    YS, XS = 120, 330
    cases = [synth.make_case(YS, XS, s, tel='ML1', os_y=20, os_x=45, n_stars=60, n_sat=2, n_cr=40) for s in (77, 78, 79)]
    case = cases[0]
    coeffs = O.xtalk_coeffs(case['xtalk'])
    d0, m0, h0, _ = R.reduce_object(ctx, dev(case['raw']), {}, 'ML1', mflat=dev(case['flat']), bpm=dev(case['bpm']),
                                    xtalk_coeffs=coeffs, exptime=60.0, ysize_chan=YS, xsize_chan=XS)
    rs = np.random.RandomState(3)
    ref = dev((d0.cpu().numpy() - 100.0 + rs.normal(0, 4, d0.shape)).astype(np.float32))
    psf = dev(bench.moffat_stamp(15, 3.5))
    sub = dict(ref=ref, ref_mask=torch.zeros_like(m0), psf_new=psf, psf_ref=psf, subimage_size=120, subimage_border=10, bkg_boxsize=30,
               cat_extract=True, trans_extract=True)
    for c in cases:
        d, m, h, hm = R.reduce_object(ctx, dev(c['raw']), {}, 'ML1', mflat=dev(case['flat']), bpm=dev(case['bpm']),
                                      xtalk_coeffs=coeffs, exptime=60.0, ysize_chan=YS, xsize_chan=XS)
        print('serial flags', {k: R.hval(h, k) for k in ('MASK-P', 'COSMIC-P', 'XTALK-P', 'SAT-P')})
        try:
            res = G.optimal_subtraction(ctx, d, new_mask=m, **sub); ctx.sync()
            print('  serial sub ok', len(res['transients']), len(res['catalog']['X_POS']))
        except Exception as e:
            print('  serial sub FAILED', e)
    geom = R.geometry(case['raw'].shape, YS, XS)
    for lanes in (1, 2):
        pipe = FramePipeline(ctx, 'ML1', geom, mflat=dev(case['flat']), bpm=dev(case['bpm']), xtalk_coeffs=coeffs, exptime=60.0,
                             pool=pool, depth=3, lanes=lanes, do_finish=True, detect_sats=True, keep_outputs=True, subtract=sub,
                             log=logging.getLogger('p'))
        pipe.run([(dev(c['raw']), {}) for c in cases], on_done=lambda i, f: print('lanes', lanes, 'frame', i, 'failed', f.failed, 'steps', f.h_out[5].numpy().tolist()))
        pipe.close()
    pool.close()


if __name__ == '__main__':
    main()
