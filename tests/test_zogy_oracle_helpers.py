"""CPU: the array-form helpers of the ZOGY oracle (used by the full-size GPU tests, where the per-pixel Python loops
of the defining functions would take minutes) against those defining functions."""
import numpy as np

import zogy_core as Z

F = np.float32


def test_psf_optflux_vec_equals_loop():
    rs = np.random.RandomState(2)
    ny, nx, S, nsrc = 60, 90, 9, 25
    img = rs.normal(0, 1, (ny, nx)).astype(F)
    V = np.abs(rs.normal(100, 10, (ny, nx))).astype(F)
    V[10:14, 20:24] = 0
    ys = rs.randint(0, ny, nsrc); xs = rs.randint(0, nx, nsrc)
    ys[:4] = [0, ny - 1, 12, 2]; xs[:4] = [0, nx - 1, 22, nx - 3]
    psfs = np.abs(rs.normal(0, 1, (nsrc, S, S))).astype(F)
    psfs /= psfs.sum(axis=(1, 2), keepdims=True)
    f0, e0 = Z.psf_optflux(img, V, psfs, ys, xs)
    f1, e1 = Z.psf_optflux_vec(img, V, psfs, ys, xs)
    np.testing.assert_allclose(f1, f0, rtol=2e-7, atol=1e-9)         # float64 sums in another order, rounded to float32
    np.testing.assert_allclose(e1, e0, rtol=2e-7)
    f2, e2 = Z.psf_optflux_vec(img, V, psfs[0], ys, xs)
    f3, e3 = Z.psf_optflux(img, V, np.repeat(psfs[:1], nsrc, 0), ys, xs)
    np.testing.assert_allclose(f2, f3, rtol=2e-7, atol=1e-9)


def test_find_transients_fast_equals_loop():
    rs = np.random.RandomState(4)
    img = rs.normal(0, 1, (200, 310)).astype(F)
    img[50:53, 60:64] = 9.0; img[51, 62] = 14.0
    img[120, 200] = -8.5; img[121, 201] = -7.0
    img[10, 10] = 6.0
    img[0, 0] = 30.0; img[199, 309] = 7.5
    img[80, 80:83] = 7.0                                   # tie: first pixel in C order
    img[150:153, 100] = 7.0; img[152, 101:104] = 7.0; img[150, 101] = -3      # L-shaped region, tie across rows
    assert Z.find_transients_fast(img, 6.0) == Z.find_transients(img, 6.0)
