"""Golden vectors for FITS tile compression (what `fpack -q Q` / `fpack` write for the
reduced float image and the uint8 mask, blackbox.py:812-857): RICE_1, one tile per row,
SUBTRACTIVE_DITHER_1.  Produced by the reference environment's astropy (its compiled CFITSIO
routines fits_quantize_float / fits_rcomp*) -- run with /opt/conda/bin/python3.9.  Inputs are
regenerated from seeds; stored: per-row compressed bytes, ZSCALE, ZZERO.
-> tests/golden/fpack.npz"""
import json
import os
import sys

import numpy as np

for _n, _f in {'asscalar': lambda a: a.item(), 'alen': len}.items():     # astropy 4.3 vs numpy 1.26
    if not hasattr(np, _n):
        setattr(np, _n, _f)
import astropy
from astropy.io import fits

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'fpack.npz')


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fpack import golden_input as make_input            # noqa: E402  (seeded inputs, plain numpy)


CASES = [dict(kind='f32', seed=1, ny=6, nx=400, q=16, dither_seed=1),
         dict(kind='f32', seed=2, ny=5, nx=1321, q=16, dither_seed=9999),      # seed wraps around N_RANDOM
         dict(kind='f32', seed=3, ny=4, nx=257, q=4, dither_seed=77),
         dict(kind='f32', seed=4, ny=3, nx=640, q=2, dither_seed=5000),
         dict(kind='u8', seed=5, ny=6, nx=300),
         dict(kind='i16', seed=6, ny=4, nx=333),
         dict(kind='i32', seed=7, ny=4, nx=200)]

out = {}
for k, c in enumerate(CASES):
    d = make_input(c['kind'], c['seed'], c['ny'], c['nx'])
    kw = dict(compression_type='RICE_1', tile_size=(c['nx'], 1))
    if c['kind'] == 'f32':
        kw.update(quantize_level=c['q'], quantize_method=1, dither_seed=c['dither_seed'])
    fits.CompImageHDU(d, **kw).writeto('/tmp/_g.fz', overwrite=True)
    with fits.open('/tmp/_g.fz', disable_image_compression=True) as hd:
        t = hd[1].data
        hdr = hd[1].header
        c['zkeys'] = {key: (hdr[key] if not isinstance(hdr[key], (bool, np.bool_)) else bool(hdr[key]))
                      for key in hdr if key.startswith('Z') and key not in ('ZHECKSUM',)}
        for r in range(c['ny']):
            out['c%d_row%d' % (k, r)] = np.frombuffer(bytes(t[r]['COMPRESSED_DATA']), np.uint8)
        if c['kind'] == 'f32':
            out['c%d_zscale' % k] = np.array([t[r]['ZSCALE'] for r in range(c['ny'])], np.float64)
            out['c%d_zzero' % k] = np.array([t[r]['ZZERO'] for r in range(c['ny'])], np.float64)
    with fits.open('/tmp/_g.fz') as hd:                                     # what a reader gets back
        out['c%d_decoded' % k] = np.asarray(hd[1].data)
out['meta'] = json.dumps(dict(cases=CASES, astropy=astropy.__version__, numpy=np.__version__), default=str)
np.savez_compressed(OUT, **out)
print('wrote', OUT, [(c['kind'], c['zkeys'].get('ZQUANTIZ'), c['zkeys'].get('ZDITHER0')) for c in CASES])
