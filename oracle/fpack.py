"""CPU restatement of FITS tile compression as fpack writes it for BlackBOX products
(blackbox.py:812-857: `fpack -q 16|4|2 -D -Y` for float images, `fpack -D -Y` for integer
ones): RICE_1, one tile per image row, floats quantised with SUBTRACTIVE_DITHER_1.

TEST INFRASTRUCTURE ONLY.  The arithmetic lives in CFITSIO (fits_rcomp*, fits_quantize_float,
FnNoise5_float, fits_init_randoms), which is not in /root/reference; it is restated here from
the published algorithm (Pence, White & Seaman 2010; FITS standard 4.0 section 10) and PINNED
by tests/golden/fpack.npz = output of the reference environment's astropy/CFITSIO
(oracle/gen_golden_fpack.py): compressed bytes, ZSCALE and ZZERO must match exactly."""
import numpy as np

N_RANDOM = 10000
N_RESERVED_VALUES = 10
NULL_VALUE = -2147483647


def fits_randoms():
    """fits_init_randoms: Park-Miller minimal standard generator, 10000 float32 values;
    the 10000th seed must be 1043618065"""
    a, m = 16807.0, 2147483647.0
    seed = 1.0
    out = np.empty(N_RANDOM, np.float32)
    for i in range(N_RANDOM):
        temp = a * seed
        seed = temp - m * int(temp / m)
        out[i] = np.float32(seed / m)
    assert int(seed) == 1043618065
    return out


_RAND = None


def randoms():
    global _RAND
    if _RAND is None:
        _RAND = fits_randoms()
    return _RAND


# --------------------------------------------------------------------------------
# Rice coding (fits_rcomp / fits_rcomp_short / fits_rcomp_byte, block size 32)
# --------------------------------------------------------------------------------
_RICE_PAR = {1: (3, 6, 8), 2: (4, 14, 16), 4: (5, 25, 32)}      # bytepix -> fsbits, fsmax, bbits


class _BitWriter:
    def __init__(self):
        self.bits = []

    def put(self, value, n):
        for k in range(n - 1, -1, -1):
            self.bits.append((int(value) >> k) & 1)

    def tobytes(self):
        b = self.bits + [0] * ((-len(self.bits)) % 8)
        return np.packbits(np.array(b, np.uint8)).tobytes() if b else b''


def rice_encode(a, bytepix, nblock=32):
    """a: 1-D integer array (int8/uint8 view as signed, int16, int32) -> bytes"""
    fsbits, fsmax, bbits = _RICE_PAR[bytepix]
    mod = 1 << (8 * bytepix)
    half = mod >> 1
    vals = [int(v) for v in np.asarray(a).astype(np.int64)]
    # values are handled as signed integers of the pixel width (uint8 data are reinterpreted)
    vals = [((v + half) % mod) - half for v in vals]
    w = _BitWriter()
    w.put(vals[0] % mod, 8 * bytepix)
    last = vals[0]
    nx = len(vals)
    for i in range(0, nx, nblock):
        blk = vals[i:i + nblock]
        diffs = []
        for v in blk:
            pd = ((v - last + half) % mod) - half                  # wraps like the C integer type
            d = (~(pd << 1)) if pd < 0 else (pd << 1)
            diffs.append(d % mod)
            last = v
        pixelsum = float(sum(diffs))
        thisblock = len(blk)
        dpsum = (pixelsum - (thisblock // 2) - 1) / thisblock
        if dpsum < 0:
            dpsum = 0.0
        psum = (int(dpsum) & 0xffffffff) >> 1
        fs = 0
        while psum > 0:
            psum >>= 1
            fs += 1
        if fs >= fsmax:
            w.put(fsmax + 1, fsbits)
            for d in diffs:
                w.put(d, bbits)
        elif fs == 0 and pixelsum == 0:
            w.put(0, fsbits)
        else:
            w.put(fs + 1, fsbits)
            for d in diffs:
                top = d >> fs
                w.put(1, top + 1)                                   # top zeros, then a one
                if fs:
                    w.put(d & ((1 << fs) - 1), fs)
    return w.tobytes()


def rice_decode(buf, nx, bytepix, nblock=32):
    """-> int64 array of nx values (signed of the pixel width)"""
    fsbits, fsmax, bbits = _RICE_PAR[bytepix]
    mod = 1 << (8 * bytepix)
    half = mod >> 1
    bits = np.unpackbits(np.frombuffer(bytes(buf), np.uint8))
    pos = 0

    def get(n):
        nonlocal pos
        v = 0
        for _ in range(n):
            v = (v << 1) | int(bits[pos])
            pos += 1
        return v
    out = np.empty(nx, np.int64)
    last = get(8 * bytepix)
    last = ((last + half) % mod) - half
    i = 0
    while i < nx:
        fs = get(fsbits) - 1
        n = min(nblock, nx - i)
        for j in range(n):
            if fs < 0:
                d = 0
            elif fs == fsmax:
                d = get(bbits)
            else:
                top = 0
                while bits[pos] == 0:
                    top += 1
                    pos += 1
                pos += 1
                d = (top << fs) | (get(fs) if fs else 0)
            pd = (d >> 1) ^ (-(d & 1))                              # undo the zig-zag mapping
            last = ((last + pd + half) % mod) - half
            out[i + j] = last
        i += n
    return out


# --------------------------------------------------------------------------------
# float quantisation (fits_quantize_float with FnNoise5_float), tile = one row, no nulls
# --------------------------------------------------------------------------------
def _lower_median(v):
    v = np.sort(np.asarray(v, np.float32))
    return v[(v.size - 1) // 2]


def noise5(row):
    """FnNoise5_float on one row without nulls -> (min, max float32; noise2, noise3, noise5 float64)"""
    v = np.asarray(row, np.float32)
    nx = v.size
    mn, mx = v.min(), v.max()
    if nx < 9:
        return mn, mx, 0.0, 0.0, 0.0
    F = np.float32
    v1, v3, v5, v7, v9 = v[0:nx - 8], v[2:nx - 6], v[4:nx - 4], v[6:nx - 2], v[8:nx]
    d2 = np.abs(v5 - v7)
    d3 = np.abs((F(2) * v5) - v3 - v7)
    d5 = np.abs((F(6) * v5) - (F(4) * v3) - (F(4) * v7) + v1 + v9)
    n2 = 1.0483579 * float(_lower_median(d2))             # doubles in CFITSIO
    n3 = 0.6052697 * float(_lower_median(d3))
    n5 = 0.1772048 * float(_lower_median(d5))
    return mn, mx, n2, n3, n5


def _nint(x):
    return int(x + 0.5) if x >= 0 else int(x - 0.5)


def quantize_row(row, irow, qlevel):
    """fits_quantize_float(row=irow (1-based tile number + ZDITHER0 - 1), ...) for one tile
    without nulls, SUBTRACTIVE_DITHER_1 -> (idata int32[nx], bscale, bzero) or None when the
    tile cannot be quantised (delta == 0 or too wide a range)"""
    v = np.asarray(row, np.float32)
    nx = v.size
    mn, mx, n2, n3, n5 = noise5(v)
    stdev = float(n3)
    if n2 != 0 and float(n2) < stdev:
        stdev = float(n2)
    if n5 != 0 and float(n5) < stdev:
        stdev = float(n5)
    delta = stdev / 4.0 if qlevel == 0 else stdev / qlevel
    if delta == 0.0:
        return None
    minval, maxval = float(mn), float(mx)
    if (maxval - minval) / delta > 2.0 * 2147483647.0 - N_RESERVED_VALUES:
        return None
    rnd = randoms()
    iseed = (irow - 1) % N_RANDOM
    nextrand = int(float(rnd[iseed]) * 500.0)
    if (maxval - minval) / delta < 2147483647.0 - N_RESERVED_VALUES:
        zeropt = minval
        iq = int(zeropt / delta + 0.5)                           # (LONGLONG) truncation
        zeropt = iq * delta
    else:
        zeropt = (minval + maxval) / 2.0
    out = np.empty(nx, np.int32)
    for i in range(nx):
        out[i] = _nint(((float(v[i]) - zeropt) / delta) + float(rnd[nextrand]) - 0.5)
        nextrand += 1
        if nextrand == N_RANDOM:
            iseed += 1
            if iseed == N_RANDOM:
                iseed = 0
            nextrand = int(float(rnd[iseed]) * 500.0)
    return out, delta, zeropt


def unquantize_row(idata, irow, bscale, bzero):
    """the reader's side (unquantize_i4r4, SUBTRACTIVE_DITHER_1): float32 values"""
    rnd = randoms()
    iseed = (irow - 1) % N_RANDOM
    nextrand = int(float(rnd[iseed]) * 500.0)
    out = np.empty(len(idata), np.float32)
    for i, q in enumerate(idata):
        out[i] = np.float32((float(q) - float(rnd[nextrand]) + 0.5) * bscale + bzero)
        nextrand += 1
        if nextrand == N_RANDOM:
            iseed += 1
            if iseed == N_RANDOM:
                iseed = 0
            nextrand = int(float(rnd[iseed]) * 500.0)
    return out


def compress_float_image(img, qlevel, dither_seed):
    """-> list of (bytes, zscale, zzero) per row"""
    res = []
    for r in range(img.shape[0]):
        q = quantize_row(img[r], r + 1 + dither_seed - 1, qlevel)
        if q is None:
            raise ValueError('row %d cannot be quantised' % r)
        idata, bscale, bzero = q
        res.append((rice_encode(idata, 4), bscale, bzero))
    return res


def golden_input(kind, seed, ny, nx):
    """seeded inputs of tests/golden/fpack.npz (RandomState only: identical under any numpy)"""
    rs = np.random.RandomState(seed)
    if kind == 'f32':
        yy, xx = np.mgrid[0:ny, 0:nx]
        d = 900 + 0.3 * xx + 2.0 * yy + rs.normal(0, 25, (ny, nx))
        for _ in range(max(1, ny * nx // 400)):
            d[rs.randint(0, ny), rs.randint(0, nx)] += rs.uniform(500, 60000)
        return d.astype(np.float32)
    if kind == 'u8':
        m = np.zeros((ny, nx), np.uint8)
        for bit, frac in ((1, 0.01), (2, 0.004), (4, 0.002), (8, 0.002), (32, 0.02), (64, 0.001)):
            m[rs.rand(ny, nx) < frac] |= bit
        return m
    if kind == 'i16':
        return (1000 + rs.normal(0, 12, (ny, nx))).astype(np.int16)
    if kind == 'i32':
        d = (rs.normal(0, 300, (ny, nx))).astype(np.int32)
        d[rs.rand(ny, nx) < 0.01] += 3000000
        return d
    raise ValueError(kind)
