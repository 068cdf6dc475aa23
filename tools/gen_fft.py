#!/usr/bin/env python
"""Generates blackbox_amd/csrc/bbx_fft_gen.h: straight-line register DFTs of small sizes for
the hand-written 2-D FFT of the ZOGY stage (bbx_zogy2.hip).

    dft<N>(float2 (&x)[N])       forward DFT, X[k] = sum_n x[n] exp(-2 pi i n k / N), natural order
                                 in and out, in place, everything in registers (full unrolling;
                                 twiddle factors are float literals rounded from float64)

Sizes: every N whose prime factors are 2, 3, 5, 7 (given on the command line / the default
list).  Decomposition: decimation in time with the largest of the radices 8, 4, 2, 3, 5, 7 that
divides N; odd prime radices use the symmetric form (x_r + x_{R-r}, x_r - x_{R-r}): (R-1)^2 / 2
real multiplications.  The inverse transform is the forward one on swapped (re, im) pairs.

Self-test: `python tools/gen_fft.py --check` compiles the header for the host with g++ and
compares every size with numpy.fft.fft.
"""
import math
import os
import subprocess
import sys
import tempfile

SIZES = [2, 3, 4, 5, 7, 8, 16]                 # the radices of bbx_zogy3.hip's plans (1400 = 5 7 5 8, 140 = 5 7 4, 128 = 8 16, ...)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'blackbox_amd', 'csrc', 'bbx_fft_gen.h')


def lit(v):
    if v == 0:
        return '0.0f'
    return repr(float(v)) + 'f'


class Emit:
    def __init__(self):
        self.lines = []
        self.n = 0

    def tmp(self):
        self.n += 1
        return 't%d' % self.n

    def add(self, s):
        self.lines.append('    ' + s)


def radix(n):
    for r in (8, 4, 2, 3, 5, 7):
        if n % r == 0:
            return r
    raise ValueError('size %d has a prime factor other than 2, 3, 5, 7' % n)


def butterfly(e, v):
    """in-place DFT of the R complex values named by v (list of (re, im) variable names);
    returns the list of output names"""
    R = len(v)
    if R == 1:
        return v
    if R == 2:
        a, b = v
        o0, o1 = (e.tmp(), e.tmp()), (e.tmp(), e.tmp())
        e.add('const float %s = %s + %s, %s = %s + %s;' % (o0[0], a[0], b[0], o0[1], a[1], b[1]))
        e.add('const float %s = %s - %s, %s = %s - %s;' % (o1[0], a[0], b[0], o1[1], a[1], b[1]))
        return [o0, o1]
    if R == 4:
        e0, e1 = butterfly(e, [v[0], v[2]])
        o0, o1 = butterfly(e, [v[1], v[3]])
        # X0 = e0 + o0, X2 = e0 - o0, X1 = e1 - i o1, X3 = e1 + i o1
        x0, x2 = butterfly(e, [e0, o0])
        x1, x3 = (e.tmp(), e.tmp()), (e.tmp(), e.tmp())
        e.add('const float %s = %s + %s, %s = %s - %s;' % (x1[0], e1[0], o1[1], x1[1], e1[1], o1[0]))
        e.add('const float %s = %s - %s, %s = %s + %s;' % (x3[0], e1[0], o1[1], x3[1], e1[1], o1[0]))
        return [x0, x1, x2, x3]
    if R == 8:
        ev = butterfly(e, [v[0], v[2], v[4], v[6]])
        od = butterfly(e, [v[1], v[3], v[5], v[7]])
        h = math.sqrt(0.5)
        tw = [od[0]]
        # od[1] * (1 - i) / sqrt2 ; od[2] * (-i) ; od[3] * (-1 - i) / sqrt2
        t1 = (e.tmp(), e.tmp())
        e.add('const float %s = (%s + %s) * %s, %s = (%s - %s) * %s;' % (t1[0], od[1][0], od[1][1], lit(h), t1[1], od[1][1], od[1][0], lit(h)))
        t2 = (od[2][1], '(-%s)' % od[2][0])
        t3 = (e.tmp(), e.tmp())
        e.add('const float %s = (%s - %s) * %s, %s = (%s + %s) * %s;' % (t3[0], od[3][1], od[3][0], lit(h), t3[1], od[3][0], od[3][1], lit(-h)))
        tw += [t1, t2, t3]
        out = [None] * 8
        for k in range(4):
            a, b = butterfly(e, [ev[k], tw[k]])
            out[k], out[k + 4] = a, b
        return out
    # odd prime: symmetric form
    half = (R - 1) // 2
    a, b = [], []
    for r in range(1, half + 1):
        ar, br = (e.tmp(), e.tmp()), (e.tmp(), e.tmp())
        e.add('const float %s = %s + %s, %s = %s + %s;' % (ar[0], v[r][0], v[R - r][0], ar[1], v[r][1], v[R - r][1]))
        e.add('const float %s = %s - %s, %s = %s - %s;' % (br[0], v[r][0], v[R - r][0], br[1], v[r][1], v[R - r][1]))
        a.append(ar); b.append(br)
    out = [None] * R
    x0 = (e.tmp(), e.tmp())
    e.add('const float %s = %s + %s, %s = %s + %s;' % (x0[0], v[0][0], ' + '.join(t[0] for t in a), x0[1], v[0][1], ' + '.join(t[1] for t in a)))
    out[0] = x0
    for k in range(1, half + 1):
        c = [math.cos(2 * math.pi * r * k / R) for r in range(1, half + 1)]
        s = [math.sin(2 * math.pi * r * k / R) for r in range(1, half + 1)]
        p, q = (e.tmp(), e.tmp()), (e.tmp(), e.tmp())
        e.add('const float %s = %s + %s, %s = %s + %s;' % (
            p[0], v[0][0], ' + '.join('%s * %s' % (lit(c[i]), a[i][0]) for i in range(half)),
            p[1], v[0][1], ' + '.join('%s * %s' % (lit(c[i]), a[i][1]) for i in range(half))))
        e.add('const float %s = %s, %s = %s;' % (
            q[0], ' + '.join('%s * %s' % (lit(s[i]), b[i][0]) for i in range(half)),
            q[1], ' + '.join('%s * %s' % (lit(s[i]), b[i][1]) for i in range(half))))
        # X_k = P - i Q ; X_{R-k} = P + i Q
        xk, xr = (e.tmp(), e.tmp()), (e.tmp(), e.tmp())
        e.add('const float %s = %s + %s, %s = %s - %s;' % (xk[0], p[0], q[1], xk[1], p[1], q[0]))
        e.add('const float %s = %s - %s, %s = %s + %s;' % (xr[0], p[0], q[1], xr[1], p[1], q[0]))
        out[k], out[R - k] = xk, xr
    return out


def dft(e, v):
    """DFT of the values named by v, natural order -> list of output names (natural order)"""
    N = len(v)
    if N == 1:
        return v
    R = radix(N)
    M = N // R
    if M == 1:
        return butterfly(e, v)
    # decimation in time: sequences x[R m + r]
    sub = [dft(e, [v[R * m + r] for m in range(M)]) for r in range(R)]
    out = [None] * N
    for k in range(M):
        u = []
        for r in range(R):
            ang = -2 * math.pi * r * k / N
            c, s = math.cos(ang), math.sin(ang)
            t = sub[r][k]
            if r * k == 0:
                u.append(t)
            else:
                w = (e.tmp(), e.tmp())
                e.add('const float %s = %s * %s - %s * %s, %s = %s * %s + %s * %s;' % (
                    w[0], t[0], lit(c), t[1], lit(s), w[1], t[0], lit(s), t[1], lit(c)))
                u.append(w)
        res = butterfly(e, u)
        for q in range(R):
            out[k + M * q] = res[q]
    return out


def gen(sizes):
    o = ['// GENERATED by tools/gen_fft.py -- do not edit.  Register DFTs (forward, natural order, in place).',
         '#pragma once', '#ifndef BBX_FFT_FN', '#define BBX_FFT_FN __device__ __forceinline__', '#endif',
         'template <int N> struct bbx_dft;', '']
    for n in sizes:
        e = Emit()
        v = [('x[%d].x' % i, 'x[%d].y' % i) for i in range(n)]
        out = dft(e, v)
        o.append('template <> struct bbx_dft<%d> {' % n)
        o.append('  static BBX_FFT_FN void run(float2 (&x)[%d]) {' % n)
        o += e.lines
        for i, t in enumerate(out):
            o.append('    x[%d].x = %s; x[%d].y = %s;' % (i, t[0], i, t[1]))
        o.append('  }')
        o.append('};')
        o.append('')
    return '\n'.join(o) + '\n'


def check(sizes):
    import numpy as np
    src = ['#include <cstdio>', '#include <cstdlib>', 'struct float2 { float x, y; };', '#define BBX_FFT_FN inline',
           '#include "bbx_fft_gen.h"', 'int main(int argc, char** argv) { int n = atoi(argv[1]);']
    for n in sizes:
        src.append('  if (n == %d) { float2 x[%d]; for (int i = 0; i < %d; i++) { if (scanf("%%f %%f", &x[i].x, &x[i].y) != 2) return 1; } '
                   'bbx_dft<%d>::run(x); for (int i = 0; i < %d; i++) printf("%%.9g %%.9g\\n", x[i].x, x[i].y); }' % (n, n, n, n, n))
    src.append('  return 0; }')
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, 'bbx_fft_gen.h'), 'w').write(gen(sizes))
        open(os.path.join(td, 't.cpp'), 'w').write('\n'.join(src))
        subprocess.check_call(['g++', '-O1', '-ffp-contract=off', '-o', os.path.join(td, 't'), os.path.join(td, 't.cpp')])
        rs = np.random.RandomState(1)
        worst = 0.0
        for n in sizes:
            x = (rs.normal(size=n) + 1j * rs.normal(size=n)).astype(np.complex64)
            inp = '\n'.join('%.9g %.9g' % (z.real, z.imag) for z in x)
            res = subprocess.run([os.path.join(td, 't'), str(n)], input=inp.encode(), stdout=subprocess.PIPE, check=True).stdout
            y = np.array([[float(a) for a in ln.split()] for ln in res.decode().strip().splitlines()])
            y = y[:, 0] + 1j * y[:, 1]
            ref = np.fft.fft(x.astype(np.complex128))
            err = np.abs(y - ref).max() / np.abs(ref).max()
            worst = max(worst, err)
            assert err < 2e-6, (n, err)
        print('register DFTs of sizes', sizes, 'agree with numpy.fft (max rel err %.1e)' % worst)


if __name__ == '__main__':
    if '--check' in sys.argv:
        check(SIZES)
    else:
        open(OUT, 'w').write(gen(SIZES))
        print('wrote', os.path.normpath(OUT))
