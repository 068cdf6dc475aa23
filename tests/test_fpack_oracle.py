"""f2 FITS tile compression: the oracle (oracle/fpack.py) against the reference environment's
CFITSIO output (tests/golden/fpack.npz, made by oracle/gen_golden_fpack.py)."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
import fpack as FP                                     # noqa: E402
make_input = FP.golden_input

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fpack.npz')


def cases():
    g = np.load(GOLD)
    return g, json.loads(str(g['meta']))['cases']


def test_random_table():
    r = FP.randoms()
    assert r.size == 10000 and r.dtype == np.float32 and 0 < r.min() and r.max() < 1


@pytest.mark.parametrize('k', [4, 5, 6])
def test_rice_integer_tiles_bytes_exact(k):
    g, cs = cases()
    c = cs[k]
    d = make_input(c['kind'], c['seed'], c['ny'], c['nx'])
    bp = {'u8': 1, 'i16': 2, 'i32': 4}[c['kind']]
    for r in range(c['ny']):
        want = g['c%d_row%d' % (k, r)].tobytes()
        assert FP.rice_encode(d[r], bp) == want, (k, r)
        back = FP.rice_decode(want, c['nx'], bp)
        assert np.array_equal(back.astype(d.dtype), d[r])


@pytest.mark.parametrize('k', [0, 1, 2, 3])
def test_float_tiles_scale_zero_bytes_exact(k):
    g, cs = cases()
    c = cs[k]
    d = make_input('f32', c['seed'], c['ny'], c['nx'])
    got = FP.compress_float_image(d, c['q'], c['dither_seed'])
    for r in range(c['ny']):
        b, zs, zz = got[r]
        assert zs == g['c%d_zscale' % k][r], (k, r, zs, g['c%d_zscale' % k][r])
        assert zz == g['c%d_zzero' % k][r], (k, r)
        assert b == g['c%d_row%d' % (k, r)].tobytes(), (k, r)
        # and the reader's view
        idata = FP.rice_decode(b, c['nx'], 4)
        back = FP.unquantize_row(idata, r + 1 + c['dither_seed'] - 1, zs, zz)
        assert np.array_equal(back, g['c%d_decoded' % k][r])
        assert np.max(np.abs(back - d[r])) <= 0.5 * zs * 1.0001 + 1e-3 * zs
