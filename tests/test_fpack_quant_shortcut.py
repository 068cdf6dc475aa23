"""The quantiser of k_fp_tile (blackbox_amd/csrc/bbx_fpack.hip) does not divide: u = fma(x - zero, 1 / delta, r), integer =
floor(u) unless fract(u) lies within 2^-17 of 0 or 1 -- then CFITSIO's own expression NINT((x - zero) / delta + r - 0.5)
(fits_quantize_float, SUBTRACTIVE_DITHER_1; oracle/fpack.py quantize_row).  This is a statement about IEEE float64
arithmetic, the same on the host: a C restatement of both expressions is compared over random and engineered inputs (values
placed on and next to the rounding ties, quotients up to 2^31), and the share of pixels that take the division is checked."""
import ctypes, os, subprocess
import numpy as np

SRC = r'''
#include <math.h>
#include <stdint.h>
static int nint_(double y) { return y >= 0. ? (int)(y + 0.5) : (int)(y - 0.5); }
/* returns the number of inputs where the filter passes and floor(u) differs from CFITSIO's integer; *nslow = inputs that take the division */
long check(const float* x, const double* zero, const double* delta, const float* r, long n, long* nslow) {
    long bad = 0, slow = 0;
    for (long i = 0; i < n; i++) {
        const double xx = (double)x[i] - zero[i], rr = (double)r[i];
        const int want = nint_((xx / delta[i]) + rr - 0.5);
        const double u = fma(xx, 1.0 / delta[i], rr);
        const double fr = u - floor(u);
        if (!(fabs(fr - 0.5) < 0.5 - 0x1p-17)) { slow++; continue; }
        if ((int)floor(u) != want) bad++;
    }
    *nslow = slow;
    return bad;
}
'''


def test_floor_of_fused_quotient_equals_cfitsio_nint(tmp_path):
    c = tmp_path / 'q.c'
    c.write_text(SRC)
    so = str(tmp_path / 'q.so')
    subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-fPIC', '-shared', '-o', so, str(c), '-lm'])
    lib = ctypes.CDLL(so)
    lib.check.restype = ctypes.c_long
    rs = np.random.RandomState(7)
    n = 4_000_000

    def run(x, zero, delta, r):
        x = np.ascontiguousarray(x, np.float32); zero = np.ascontiguousarray(zero, np.float64)
        delta = np.ascontiguousarray(delta, np.float64); r = np.ascontiguousarray(r, np.float32)
        ns = ctypes.c_long(0)
        bad = lib.check(x.ctypes.data_as(ctypes.c_void_p), zero.ctypes.data_as(ctypes.c_void_p), delta.ctypes.data_as(ctypes.c_void_p),
                        r.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(len(x)), ctypes.byref(ns))
        return bad, ns.value

    r = rs.random_sample(n).astype(np.float32)                        # the dither table holds float32 values in [0, 1)
    # (1) a reduced image's numbers: sky ~ 300 +- 9, delta = noise / q, zero = a multiple of delta below the minimum
    delta = (9.0 / rs.choice([2, 4, 16], n)) * (1 + 0.1 * rs.standard_normal(n))
    zero = np.floor((300 - 60 * rs.random_sample(n)) / delta) * delta
    x = 300 + 9 * rs.standard_normal(n)
    bad, slow = run(x, zero, delta, r)
    assert bad == 0 and slow < 1e-4 * n                                # (2 x 2^-17 = 1.5e-5 of the pixels divide)
    # (2) engineered: x on the lattice of ties / integers of u, and a few float32 ulps next to it
    m = rs.randint(-1000, 100000, n)
    for off in (0.0, 0.5):
        for eps in (0.0, 1e-9, -1e-9, 3e-7, -3e-7, 1e-5, -1e-5):
            xe = zero + delta * (m + off - r.astype(np.float64) + eps)
            bad, slow = run(xe, zero, delta, r)
            assert bad == 0, (off, eps, bad)
    # (3) quotients up to 2^31 (CFITSIO's range check lets (max - min) / delta reach 2 x 2^31 around a mid-range zero point)
    delta = 10.0 ** rs.uniform(-6, 2, n)
    zero = rs.uniform(-1e3, 1e3, n)
    t = rs.uniform(-2.0 ** 31, 2.0 ** 31, n)
    bad, slow = run(zero + t * delta, zero, delta, r)
    assert bad == 0
    t = np.rint(rs.uniform(-2.0 ** 31, 2.0 ** 31, n)) + rs.choice([0.0, 0.5, 0.5 - 2.0 ** -20, 0.5 + 2.0 ** -20], n)
    bad, slow = run((zero + (t - r) * delta).astype(np.float64), zero, delta, r)
    assert bad == 0
