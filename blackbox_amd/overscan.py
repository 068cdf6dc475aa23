"""Host side of os_corr: the float64 fits on the small overscan vectors.

The bulk strip reductions run on the GPU (csrc/bbx_overscan.hip); what is left
are <= 5300-point polynomial fits and a 180-point smoothing spline per channel
(blackbox.py:6497-6517 and 6660-6814), which SURVEY.md section 7 keeps on the
host in numpy/scipy because the reference itself uses np.polyfit (LAPACK lstsq)
and scipy's FITPACK spline -- re-deriving those bit for bit would be pointless.

Numerical convention of the horizontal-overscan statistics: the reference
environment (astropy 4.3 + bottleneck) accumulates the clipped mean/std of the
float32 strips in float32 with a running sum.  ``accum='f32seq'`` (default)
reproduces that order, which makes the reduced pixels bit-identical to the
golden vectors; ``accum='f64'`` uses float64 accumulators (differences of <= 2
float32 ulp per pixel).
"""
import warnings

import numpy as np
from scipy import interpolate, ndimage

# optional C versions of the two hottest helpers (blackbox_amd/chost/bbx_host.c, built by
# `make`): the same float operations in the same order, ~10x less interpreter overhead.
# Without the library the numpy code below runs -- both are held to each other bit for bit
# by tests/test_host_overscan.py.
_HOST = None
try:
    import ctypes as _C
    import os as _os
    _HOST = _C.CDLL(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'libbbx_host.so'))
    _HOST.bbx_hos_column_stats_f32seq.restype = _C.c_int
    _HOST.bbx_hos_column_stats_f32seq.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_int, _C.c_int, _C.c_void_p, _C.c_void_p,
                                                  _C.c_void_p]
    _HOST.bbx_clipped_stats_flat_f32seq.restype = _C.c_int64
    _HOST.bbx_clipped_stats_flat_f32seq.argtypes = [_C.c_void_p, _C.c_int64, _C.c_double, _C.c_int, _C.c_void_p]
    _HOST.bbx_channel_solve_ml1.restype = _C.c_int
    _HOST.bbx_channel_solve_ml1.argtypes = ([_C.c_int, _C.c_void_p, _C.c_int, _C.c_void_p, _C.c_int, _C.c_int, _C.c_int,
                                             _C.c_int, _C.c_int, _C.c_double, _C.c_void_p, _C.c_void_p, _C.c_void_p]
                                            + [_C.c_void_p] * 5)
    _HOST.bbx_lstsq_direct.restype = _C.c_int
    _HOST.bbx_lstsq_direct.argtypes = [_C.c_void_p, _C.c_int64, _C.c_int, _C.c_void_p, _C.c_double, _C.c_void_p, _C.c_void_p]
    _HOST.bbx_host_set_dgelsd.argtypes = [_C.c_void_p]
    _HOST.bbx_polyfit_prep.restype = _C.c_int64
    _HOST.bbx_polyfit_prep.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_int64, _C.c_int, _C.c_void_p, _C.c_void_p]
except OSError:
    _HOST = None


def _install_numpy_lapack():
    """hand the C driver the dgelsd of the BLAS numpy itself is linked with (the ILP64 OpenBLAS that the
    numpy wheels bundle: `scipy_dgelsd_64_`), so that np.linalg.lstsq's arithmetic is reached without
    the interpreter; any other numpy build keeps the callback (same results, slower)"""
    if _HOST is None:
        return False
    try:
        import glob
        import numpy.linalg._umath_linalg                       # noqa: F401  (makes sure the library is mapped)
        libs = glob.glob(_os.path.join(_os.path.dirname(_os.path.dirname(np.__file__)), 'numpy.libs', 'libscipy_openblas64_*.so'))
        if len(libs) != 1:
            return False
        blas = _C.CDLL(libs[0])                                  # already mapped by numpy: the same instance
        fn = _C.cast(blas.scipy_dgelsd_64_, _C.c_void_p).value
        if not fn:
            return False
        _HOST.bbx_host_set_dgelsd(fn)
        # the symbol name says ILP64, but trust a measurement: one small system solved through the installed entry must
        # reproduce np.linalg.lstsq bit for bit (a build with another integer width or routine would not)
        rs = np.random.RandomState(0)
        lhs = np.ascontiguousarray(rs.normal(size=(40, 5)))
        rhs = np.ascontiguousarray(rs.normal(size=40))
        coef, rank = np.empty(5), _C.c_int(0)
        rc = _HOST.bbx_lstsq_direct(lhs.ctypes.data, 40, 5, rhs.ctypes.data, float(40 * np.finfo(np.float64).eps),
                                    coef.ctypes.data, _C.addressof(rank))
        want = np.linalg.lstsq(lhs, rhs, rcond=40 * np.finfo(np.float64).eps)
        if rc != 0 or rank.value != int(want[2]) or not np.array_equal(coef, want[0]):
            _HOST.bbx_host_set_dgelsd(None)
            return False
        return True
    except (OSError, AttributeError, ImportError):
        return False


DIRECT_LAPACK = _install_numpy_lapack()

USE_DIRECT_LAPACK = True   # the C driver calls numpy's own dgelsd itself instead of np.linalg.lstsq through a callback
USE_C_DRIVER = True   # channel_solve through libbbx_host.so (bit-identical; tests switch it off to compare)
IDX_SWITCH = 150      # blackbox.py:6683
OVERLAP = 30          # blackbox.py:6684


def _seqsum32(v):
    return np.cumsum(v, dtype=np.float32)[-1] if v.size else np.float32(0)


def _mean_std_flat(v, accum):
    n = v.size
    if accum == 'f32seq' and v.dtype == np.float32:
        mean = np.float32(_seqsum32(v) / np.float32(n))
        dev = v - mean
        std = np.float32(np.sqrt(_seqsum32(dev * dev) / np.float32(n)))
        return mean, std
    v = v.astype(np.float64)
    mean = v.sum() / n
    return mean, np.sqrt(((v - mean) ** 2).sum() / n)


def clipped_stats_flat(values, sigma=3.0, maxiters=5, accum='f32seq'):
    """astropy sigma_clipped_stats(values, sigma=sigma, cenfunc='mean') on a
    flattened array -> (mean, std, n_survivors)"""
    v = np.ascontiguousarray(values).ravel()
    if _HOST is not None and accum == 'f32seq' and v.dtype == np.float32:
        out = np.empty(2)
        m = _HOST.bbx_clipped_stats_flat_f32seq(v.ctypes.data, v.size, float(sigma), int(maxiters), out.ctypes.data)
        if m < 0:
            raise MemoryError('bbx_clipped_stats_flat_f32seq')
        if m == 0:
            return np.nan, np.nan, 0
        return np.float32(out[0]), np.float32(out[1]), int(m)
    v = v[np.isfinite(v)]
    for _ in range(maxiters):
        if v.size == 0:
            break
        mean, std = _mean_std_flat(v, accum)
        lo = float(mean) - float(std) * sigma
        hi = float(mean) + float(std) * sigma
        if accum == 'f32seq' and v.dtype == np.float32:
            keep = (v >= np.float32(lo)) & (v <= np.float32(hi))
        else:
            vv = v.astype(np.float64)
            keep = (vv >= lo) & (vv <= hi)
        if keep.all():
            break
        v = v[keep]
    if v.size == 0:
        return np.nan, np.nan, 0
    mean, std = _mean_std_flat(v, accum)
    return mean, std, v.size


_VANDER = {}
_VANDER_LOCK = __import__('threading').Lock()


def _vander_full(start, n, order):
    """np.vander(np.arange(start, start + n) + 0.0, order), cached: np.polyfit builds it from
    the selected abscissae on every call, row by row (cumulative products), so the rows of
    the full matrix picked by the fit mask are the identical numbers"""
    key = (start, n, order)
    v = _VANDER.get(key)
    if v is None:
        # (callers hand the array's address to C and rely on the cache to keep it alive: one array per key, ever --
        # several threads may ask for a new key at once)
        with _VANDER_LOCK:
            v = _VANDER.get(key)
            if v is None:
                v = _VANDER[key] = np.vander(np.arange(start, start + n) + 0.0, order)
    return v


def polyfit_exact(start, n, mask, y, deg):
    """np.polyfit(np.arange(start, start+n)[mask], y, deg) -- the same float operations in the
    same order (scaled Vandermonde, LAPACK gelsd through np.linalg.lstsq), without rebuilding
    the Vandermonde matrix and without np.polyfit's argument handling.
    -> (coefficients high -> low order, rank)"""
    order = deg + 1
    V = _vander_full(start, n, order)
    rhs = np.asarray(y) + 0.0
    if _HOST is not None and rhs.size > 0:
        mk = np.ascontiguousarray(mask, dtype=np.uint8)
        lhs = np.empty((rhs.size, order))
        scale = np.empty(order)
        m = _HOST.bbx_polyfit_prep(V.ctypes.data, mk.ctypes.data, n, order, lhs.ctypes.data, scale.ctypes.data)
        if m != rhs.size:
            raise ValueError('polyfit_exact: {} ordinates for {} selected abscissae'.format(rhs.size, m))
    else:
        lhs = V[mask]
        scale = np.sqrt((lhs * lhs).sum(axis=0))
        lhs /= scale
    rcond = lhs.shape[0] * np.finfo(np.float64).eps
    c, _, rank, _ = np.linalg.lstsq(lhs, rhs, rcond)
    c = (c.T / scale).T
    return c, rank


def vos_polyfit(mean_vos_col, nrows, i_chan, poldeg=3):
    """blackbox.py:6497-6556.  -> (fit[dy] float64, coeffs low->high order,
    polyfit_ok, mean level)"""
    nrows_chan = mean_vos_col.size
    y_vos = np.arange(nrows_chan)
    polyfit_ok = True
    p = None
    try:
        mean, stddev, _ = clipped_stats_flat(mean_vos_col, sigma=5, accum='f64')
        if stddev == 0:
            mask_fit = np.ones(nrows_chan, dtype=bool)
        else:
            with np.errstate(invalid='ignore'):
                mask_fit = np.abs(mean_vos_col - mean) / stddev <= 5
        if i_chan < 8:
            mask_fit[nrows:] = False
        else:
            mask_fit[:nrows_chan - nrows] = False
        # os_corr runs with warnings as errors (6432): a rank-deficient fit (np.polyfit's
        # RankWarning) counts as a failed fit
        p, rank = polyfit_exact(0, nrows_chan, mask_fit, mean_vos_col[mask_fit], poldeg)
        if rank != poldeg + 1:
            p = None
            raise ValueError('rank-deficient vertical-overscan fit')
    except Exception:
        polyfit_ok = False
    if p is None:
        # the reference would reuse the previous channel's coefficients (or raise on
        # the first channel); treat as a failed fit with the median level
        level = np.nanmedian(mean_vos_col)
        return np.full(nrows_chan, level), np.full(poldeg + 1, np.nan), False, level
    fit = np.polyval(p, y_vos)
    if not np.all(np.isfinite(fit)):
        polyfit_ok = False
    if polyfit_ok:
        level = np.mean(fit)
    else:
        level = np.nanmedian(mean_vos_col)
        fit = np.full(nrows_chan, level)
    return fit, p[::-1], polyfit_ok, level


def _open2(m):
    """ndimage.binary_opening(m, structure=np.ones(2)) of a 1-D bool array: a True survives
    iff its left or right neighbour is True (outside = False)"""
    left = np.zeros_like(m); left[1:] = m[:-1]
    right = np.zeros_like(m); right[:-1] = m[1:]
    return m & (left | right)


def _dilate5x5(m):
    """ndimage.binary_dilation(m, structure=np.ones((3, 3)), iterations=2) of a 2-D bool
    array = OR over the 5x5 neighbourhood (outside = False)"""
    ny, nx = m.shape
    p = np.zeros((ny + 4, nx + 4), bool)
    p[2:-2, 2:-2] = m
    rows = p[:, 0:nx] | p[:, 1:nx + 1] | p[:, 2:nx + 2] | p[:, 3:nx + 3] | p[:, 4:nx + 4]
    return rows[0:ny] | rows[1:ny + 1] | rows[2:ny + 2] | rows[3:ny + 3] | rows[4:ny + 4]


def hos_mask_ml1(data_hos, data_limit=2000):
    """blackbox.py:6586-6614 (the two scipy.ndimage morphology calls written out for these
    tiny arrays; tests/test_host_overscan.py holds them against scipy)"""
    mask_hos = data_hos > data_limit
    if not mask_hos.any():
        return mask_hos
    mask_x = np.sum(mask_hos, axis=0) > 0.5 * mask_hos.shape[0]
    mask_x_open = _open2(mask_x)
    mask_hos[:, np.logical_xor(mask_x, mask_x_open)] = False
    return _dilate5x5(mask_hos)


def hos_column_stats(data_hos, mask_hos, accum='f32seq'):
    """blackbox.py:6649-6659: sigma_clip(axis=0, sigma=2.5, cenfunc='mean') then
    per-column count / mean / std(ddof=1).  float32 results like the reference."""
    if _HOST is not None and accum == 'f32seq' and data_hos.dtype == np.float32:
        d = np.ascontiguousarray(data_hos)
        mk = np.ascontiguousarray(mask_hos, dtype=np.uint8)
        nrow, ncol = d.shape
        n = np.empty(ncol, np.int64); mean = np.empty(ncol, np.float32); std = np.empty(ncol, np.float32)
        if _HOST.bbx_hos_column_stats_f32seq(d.ctypes.data, mk.ctypes.data, nrow, ncol, n.ctypes.data, mean.ctypes.data,
                                              std.ctypes.data) != 0:
            raise MemoryError('bbx_hos_column_stats_f32seq')
        return n, mean, std
    d64 = data_hos.astype(np.float64)
    ok = np.isfinite(d64) & ~mask_hos
    lo = np.full(d64.shape[1], np.nan)
    hi = np.full(d64.shape[1], np.nan)
    cur = ok.copy()
    with np.errstate(invalid='ignore', divide='ignore'):
        for _ in range(5):
            n = cur.sum(axis=0)
            mean = np.where(cur, d64, 0.0).sum(axis=0) / n
            dev = np.where(cur, mean[None, :] - d64, 0.0)
            std = np.sqrt((dev * dev).sum(axis=0) / n)
            upd = n > 0
            lo = np.where(upd, mean - 2.5 * std, lo)
            hi = np.where(upd, mean + 2.5 * std, hi)
            cur = cur & (d64 >= lo[None, :]) & (d64 <= hi[None, :])
        ok = ok & ~(d64 < lo[None, :]) & ~(d64 > hi[None, :])
        n = ok.sum(axis=0)
        if accum == 'f32seq':
            f = np.float32
            tot = np.zeros(d64.shape[1], f)
            for i in range(data_hos.shape[0]):
                tot = tot + np.where(ok[i], data_hos[i], f(0))
            mean = tot / n.astype(f)
            dev = np.where(ok, data_hos - mean[None, :], f(0)).astype(f)
            sq = dev * dev
            tot2 = np.zeros_like(tot)
            for i in range(data_hos.shape[0]):
                tot2 = tot2 + sq[i]
            std = np.sqrt(tot2 / (n - 1).astype(f))
        else:
            mean = np.where(ok, d64, 0.0).sum(axis=0) / n
            dev = np.where(ok, d64 - mean[None, :], 0.0)
            std = np.sqrt((dev * dev).sum(axis=0) / (n - 1))
            mean = mean.astype(np.float32)
            std = std.astype(np.float32)
    return n, mean, std


def _polyfit_quiet(x, y, deg):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return np.polyfit(x, y, deg)


def _running_median3(y):
    """[np.median(y[max(k-1,3):min(k+2,n)]) for k in range(3, n)] as float32, vectorised
    (blackbox.py:6703-6708): interior points take the median of 3 neighbours, the two
    end points the mean of 2 (np.median of an even count)."""
    n = y.size
    if n < 6:
        return np.asarray([np.median(y[max(k - 1, 3):min(k + 2, n)]) for k in range(3, n)], np.float32)
    a, b, c = y[3:n - 2], y[4:n - 1], y[5:n]
    mid = np.maximum(np.minimum(a, b), np.minimum(np.maximum(a, b), c))
    out = np.empty(n - 3, np.float32)
    out[0] = (y[3] + y[4]) / np.float32(2)
    out[1:-1] = mid
    out[-1] = (y[n - 2] + y[n - 1]) / np.float32(2)
    return out


def hos_fit(n, mean_hos, std_hos, mask_sat_row=None, bg2_chan9=False,
            accum='f32seq', lazy_spline=True):
    """blackbox.py:6660-6814 -> oscan float64[ncols]"""
    ncols = mean_hos.size
    mask_valid = n > 1
    xcol = np.arange(ncols) + 1
    err_hos = np.zeros(ncols, np.float32)
    weights = np.zeros(ncols, np.float32)
    with np.errstate(invalid='ignore', divide='ignore'):
        err_hos[mask_valid] = (std_hos[mask_valid].astype(np.float64) /
                               np.sqrt(n[mask_valid])).astype(np.float32)
        mask_nonzero = err_hos != 0
        weights[mask_nonzero] = np.float32(1) / err_hos[mask_nonzero]
    if np.all(mask_valid[0:3]):
        weights[0:3] = 0
    idx_fit = np.arange(IDX_SWITCH + OVERLAP)
    npoints = int(np.sum(mask_valid[idx_fit] & mask_nonzero[idx_fit]))
    sel = mask_valid[idx_fit]
    y2fit = mean_hos[idx_fit][sel].copy()
    nfit = y2fit.size
    # columns below IDX_SWITCH take the raw column mean wherever it is usable; the spline
    # value only survives in the others.  The reference always builds the spline; here it is
    # built only when such a column exists (same output, the spline fit is the costly part).
    mask_usemean = mask_valid.copy()
    if mask_sat_row is not None:
        mask_usemean &= ~mask_sat_row
    mask_usemean[IDX_SWITCH:] = False
    keeps_spline = ~mask_usemean[:IDX_SWITCH]
    keeps_spline[0:3] &= ~mask_valid[0:3]
    splfit = None
    if lazy_spline is False or keeps_spline.any():
        # 3-point running median of the points to fit (6703-6708), from the original values
        y2fit[3:] = _running_median3(y2fit)
        xs, ws = xcol[idx_fit][sel], weights[idx_fit][sel]
        with warnings.catch_warnings():
            warnings.simplefilter('error')
            try:
                splfit = interpolate.UnivariateSpline(xs, y2fit, w=ws, k=2, s=npoints)
            except UserWarning:
                warnings.simplefilter('ignore')
                splfit = interpolate.UnivariateSpline(xs, y2fit, w=ws, k=3, s=1.5 * npoints)
    mask_valid_poly = mask_valid.copy()
    mask_valid_poly[0:IDX_SWITCH - OVERLAP] = False
    mhp = mean_hos[mask_valid_poly]
    mean, stddev, _ = clipped_stats_flat(mhp, sigma=5, accum=accum)
    if stddev == 0:
        keep = np.ones(mhp.size, dtype=bool)
    elif accum == 'f32seq':
        with np.errstate(invalid='ignore'):
            keep = np.abs(mhp - np.float32(mean)) / np.float32(stddev) <= np.float32(5)
    else:
        with np.errstate(invalid='ignore'):
            keep = np.abs(mhp.astype(np.float64) - mean) / stddev <= 5
    mask_valid_poly[mask_valid_poly] = keep
    err3 = np.float32(3) * err_hos

    def fit_iter(mask_fit, deg):
        fit = None
        for _ in range(3):
            p, _ = polyfit_exact(1, ncols, mask_fit, mean_hos[mask_fit], deg)
            fit = np.polyval(p, xcol)
            with np.errstate(invalid='ignore'):
                new_mask = mask_fit & (np.abs(fit - mean_hos) <= err3)
            if np.array_equal(new_mask, mask_fit):
                break                       # the same points again: the next fit would be this one
            mask_fit[:] = new_mask
        return fit

    if not bg2_chan9:
        oscan = fit_iter(mask_valid_poly, 7)
    else:
        idx_split = 654                                   # blackbox.py:6763
        m1 = mask_valid_poly.copy()
        m1[idx_split:] = False
        fit1 = fit_iter(m1, 5)
        m2 = mask_valid_poly.copy()
        m2[:idx_split] = False
        fit2 = fit_iter(m2, 5)
        oscan = fit1
        oscan[idx_split:] = fit2[idx_split:]
    if splfit is not None:
        oscan[0:IDX_SWITCH] = splfit(xcol[0:IDX_SWITCH])
    oscan[0:3][mask_valid[0:3]] = mean_hos[0:3][mask_valid[0:3]]
    oscan[mask_usemean] = mean_hos[mask_usemean]
    return oscan


# --------------------------------------------------------------------------------
# per-channel work units (what the host worker pool executes; pure numpy/scipy,
# picklable arguments, no GPU)
# --------------------------------------------------------------------------------
class OverscanFailure(Exception):
    """os_corr raised while it was working on channel [chan] (0-based).  Inside the reference's
    os_corr every warning is an error (blackbox.py:6432), and it rewrites the frame channel by
    channel IN PLACE; when it raises, blackbox_reduce crops the data sections out of that
    half-processed array (1541-1585).  So the pixels depend on how far it got: channels before
    [chan] carry both overscan corrections, channel [chan] its vertical-overscan fit [fit]
    (the vertical step itself never raises: 6503-6556 fall back to the nanmedian), the others
    nothing.  args = (chan, fit, coeffs, ok, message): picklable, crosses the worker pool."""

    def __init__(self, chan, fit, coeffs, ok, message):
        Exception.__init__(self, chan, fit, coeffs, ok, message)
        self.chan, self.fit, self.coeffs, self.ok, self.message = chan, fit, coeffs, ok, message

    def __str__(self):
        return 'channel {}: {}'.format(self.chan + 1, self.message)


def channel_phase1(c, mean_vos_col, hos, ysz, xsz, poldeg=3, accum='f32seq'):
    """vertical-overscan fit and horizontal-overscan level of channel [c].
    [hos]: gain-corrected overscan rows (hos_rows, dx) float32 as copied from the device.
    -> dict(fit, coeffs, ok, level, dlevel, strip)"""
    dy = mean_vos_col.size
    hos_rows = hos.shape[0]
    fit, coeffs, ok, level = vos_polyfit(mean_vos_col, ysz, c, poldeg)
    # overscan rows after the vertical fit: float32 - float64 -> float32 (blackbox.py:6553)
    rl0 = (dy - hos_rows) if c < 8 else 0
    strip = (hos.astype(np.float64) - fit[rl0:rl0 + hos_rows, None]).astype(np.float32)
    window = strip[:, xsz - 300:xsz]                      # blackbox.py:6565-6566 (python slice on the dx-wide strip)
    dlevel = clipped_stats_flat(window, accum=accum)[0] if window.size else np.nan
    if not np.isfinite(dlevel):
        # the reference subtracts the NaN level from the strip, which overlaps the vertical-overscan
        # section; the clipped statistics of that section (6572) then warn = raise
        raise OverscanFailure(c, fit, coeffs, ok, 'level of the horizontal overscan is not finite (window of {} columns)'
                              .format(window.shape[1]))
    strip -= np.float32(dlevel)
    return dict(fit=fit, coeffs=coeffs, ok=ok, level=level, dlevel=float(dlevel), strip=strip)


def channel_phase2(c, strip, xsz, tel, data_limit=2000, mask_sat_row=None, accum='f32seq'):
    """horizontal-overscan vector of channel [c] -> oscan float64[xsz]"""
    data_hos = strip[:, :xsz]
    if tel == 'ML1':
        mask_hos = hos_mask_ml1(data_hos, data_limit)
        mask_sat_row = None
    else:
        mask_hos = np.zeros(data_hos.shape, dtype=bool) | mask_sat_row[None, :]
    n, mean_hos, std_hos = hos_column_stats(data_hos, mask_hos, accum=accum)
    return hos_fit(n, mean_hos, std_hos, mask_sat_row, bg2_chan9=(tel == 'BG2' and c == 8), accum=accum)


def _lstsq_cb(lhs_p, m, order, rhs_p, rcond, coef_p, rank_p):
    """np.linalg.lstsq for the C driver (bbx_channel_solve_ml1): views on its buffers, no copies"""
    try:
        lhs = np.frombuffer((_C.c_char * (8 * m * order)).from_address(lhs_p), np.float64).reshape(m, order)
        rhs = np.frombuffer((_C.c_char * (8 * m)).from_address(rhs_p), np.float64)
        c, _, rank, _ = np.linalg.lstsq(lhs, rhs, rcond)
        np.frombuffer((_C.c_char * (8 * order)).from_address(coef_p), np.float64)[:] = c
        _C.c_int.from_address(rank_p).value = int(rank)
        return 0
    except Exception:
        return 1


_LSTSQ_CB = _C.CFUNCTYPE(_C.c_int, _C.c_void_p, _C.c_int64, _C.c_int, _C.c_void_p, _C.c_double, _C.c_void_p,
                         _C.c_void_p)(_lstsq_cb)


def _channel_solve_c(c, mean_vos_col, hos, ysz, xsz, poldeg, data_limit):
    """the whole channel in the C helper (LAPACK through the callback); None when a situation
    turns up that only the numpy path handles (the helper says which)"""
    col = np.ascontiguousarray(mean_vos_col, np.float64)
    h = np.ascontiguousarray(hos, np.float32)
    dy, (hos_rows, dx) = col.size, h.shape
    fit, coeffs, oscan = np.empty(dy), np.empty(poldeg + 1), np.empty(xsz)
    level, dlevel = _C.c_double(), _C.c_double()
    rc = _HOST.bbx_channel_solve_ml1(c, col.ctypes.data, dy, h.ctypes.data, hos_rows, dx, ysz, xsz, poldeg, float(data_limit),
                                     _vander_full(0, dy, poldeg + 1).ctypes.data, _vander_full(1, xsz, 8).ctypes.data,
                                     None if (DIRECT_LAPACK and USE_DIRECT_LAPACK) else _LSTSQ_CB, fit.ctypes.data, coeffs.ctypes.data,
                                     _C.addressof(level),
                                     _C.addressof(dlevel), oscan.ctypes.data)
    if rc < 0:
        raise MemoryError('bbx_channel_solve_ml1')
    if rc != 0:
        return None
    return dict(fit=fit, coeffs=coeffs, ok=True, level=level.value, dlevel=dlevel.value, oscan=oscan)


def fast_path_available(accum, hos_dtype):
    """the per-channel C driver (bbx_channel_solve_ml1) applies: a channel is one call that runs without the interpreter lock"""
    return _HOST is not None and USE_C_DRIVER and accum == 'f32seq' and hos_dtype == np.float32


def channel_solve(args):
    """both phases for telescopes without the saturated-column step (ML1)"""
    c, mean_vos_col, hos, ysz, xsz, poldeg, tel, data_limit, accum = args
    if (_HOST is not None and USE_C_DRIVER and tel == 'ML1' and accum == 'f32seq' and hos.dtype == np.float32
            and not (tel == 'BG2' and c == 8)):
        r = _channel_solve_c(c, mean_vos_col, hos, ysz, xsz, poldeg, data_limit)
        if r is not None:
            return r
    r = channel_phase1(c, mean_vos_col, hos, ysz, xsz, poldeg, accum)
    try:
        r['oscan'] = channel_phase2(c, r.pop('strip'), xsz, tel, data_limit, None, accum)
    except Exception as e:
        raise OverscanFailure(c, r['fit'], r['coeffs'], r['ok'], 'horizontal overscan: {}: {}'.format(type(e).__name__, e))
    return r


def _phase1_star(args):
    return channel_phase1(*args)


def _phase2_star(args):
    return channel_phase2(*args)
