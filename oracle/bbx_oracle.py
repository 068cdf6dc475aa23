"""TEST INFRASTRUCTURE -- CPU restatement (numpy/scipy) of the reference hot path.

NOT part of the product: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.  The product path (blackbox_amd/) never
does and has no CPU fallback.

Every function cites the reference lines (``/root/reference/blackbox.py`` unless
another file is named) it restates.  Pinning status:

* calibration / masks / crosstalk / header counts (define_sections, gain_corr,
  os_corr, mask_init, fill_sat_holes, xtalk_corr, mask_header, edge fill):
  PINNED -- tests/test_oracle_golden.py checks them against tests/golden/*.npz,
  which were produced by running the reference's own functions
  (oracle/gen_golden.py).
* LA-Cosmic (astroscrappy), satellite trails (acstools), background mesh and
  ZOGY: the arithmetic lives in third-party packages that are absent from
  /root/reference and from this container => PARITY UNPINNED; restated from the
  published algorithms (see oracle/lacosmic.py etc.) and defended by property
  tests.

numpy semantics: the golden vectors were made with numpy 1.26 (legacy
value-based casting) + astropy 4.3.1 + bottleneck 1.3.2; this module runs under
numpy 2.2 (NEP 50).  Every mixed-precision operation is therefore written with
explicit casts so that it means the same thing under both.

``accum``: the reference's clipped statistics of float32 strips are accumulated
by bottleneck in float32 with a naive running sum (relative error ~1e-6..1e-5).
accum='f64' (default) computes the same statistics with float64 accumulators
(the mathematically intended value; this is what the HIP path is held to,
bit for bit); accum='bn32' mimics bottleneck's float32 running sums and is used
only to show that the restatement reproduces the golden vectors exactly.
"""
import warnings

import numpy as np
from scipy import interpolate, ndimage

NY, NX = 2, 8
MASK_VALUE = {'bad': 1, 'cosmic ray': 2, 'saturated': 4,
              'saturated-connected': 8, 'satellite trail': 16, 'edge': 32,
              'crosstalk': 64}


# --------------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------------
def define_sections(data_shape, ysize_chan, xsize_chan, xbin=1, ybin=1):
    """blackbox.py:6334-6402.  Returns chan_sec, data_sec, os_sec_hori,
    os_sec_vert, data_sec_red as tuples of 16 (slice_y, slice_x)."""
    ysize, xsize = data_shape
    dy, dx = ysize // NY, xsize // NX
    ysize_chan //= ybin
    xsize_chan //= xbin
    ysize_os = (ysize - NY * ysize_chan) // NY
    xsize_os = (xsize - NX * xsize_chan) // NX
    chan_sec = tuple((slice(y, y + dy), slice(x, x + dx))
                     for y in range(0, ysize, dy) for x in range(0, xsize, dx))
    data_sec = tuple((slice(y, y + ysize_chan), slice(x, x + xsize_chan))
                     for y in range(0, ysize, dy + ysize_os)
                     for x in range(0, xsize, dx))
    ncut_vert = max(5 // xbin, 1)
    os_sec_vert = tuple((slice(y, y + dy),
                         slice(x + xsize_chan + ncut_vert, x + dx - 1))
                        for y in range(0, ysize, dy) for x in range(0, xsize, dx))
    ncut_hori = max(10 // ybin, 1)
    cut = ysize_os - ncut_hori
    os_sec_hori = tuple((slice(y, y + cut), slice(x, x + dx))
                        for y in range(dy - cut, dy + cut, cut)
                        for x in range(0, xsize, dx))
    data_sec_red = tuple((slice(y, y + ysize_chan), slice(x, x + xsize_chan))
                         for y in range(0, ysize - NY * ysize_os, ysize_chan)
                         for x in range(0, xsize - NX * xsize_os, xsize_chan))
    return chan_sec, data_sec, os_sec_hori, os_sec_vert, data_sec_red


# --------------------------------------------------------------------------------
# sigma clipping (astropy 4.3.1 semantics; Appendix B of SURVEY.md)
# --------------------------------------------------------------------------------
def _clip_bounds_rows(buf, valid, sigma, maxiters=5):
    """astropy.stats._fast_sigma_clip (C gufunc), cenfunc='mean', stdfunc='std':
    per row of [buf] (float64, shape (n, m)) iterate mean/std over the surviving
    values (std with ddof=0), bounds mean -/+ sigma*std, survivors are those
    inside the closed interval; stop when nothing was removed or after
    [maxiters] bound computations.  Returns (lo, hi) float64 arrays."""
    n = buf.shape[0]
    lo = np.full(n, np.nan)
    hi = np.full(n, np.nan)
    for i in range(n):
        v = buf[i][valid[i]]
        it = 0
        while v.size > 0:
            mean = v.sum() / v.size
            std = np.sqrt(((mean - v) ** 2).sum() / v.size)
            lo[i] = mean - sigma * std
            hi[i] = mean + sigma * std
            keep = (v >= lo[i]) & (v <= hi[i])
            nkeep = int(keep.sum())
            if nkeep == v.size:
                break
            v = v[keep]
            it += 1
            if it >= maxiters:
                break
    return lo, hi


def sigma_clip_axis(data, axis, sigma, mask=None, mask_value=None, maxiters=5):
    """SigmaClip._sigmaclip_fast (sigma_clipping.py:305-375): returns the
    boolean mask of rejected elements (input mask | non-finite | outside the
    final bounds) in the shape of [data]."""
    d = np.asarray(data)
    dm = np.moveaxis(d, axis, -1)
    bad = ~np.isfinite(dm)
    if mask is not None:
        bad = bad | np.moveaxis(np.asarray(mask, bool), axis, -1)
    if mask_value is not None:
        # np.ma.masked_values(data, v): |x - v| <= atol + rtol*|v| with
        # rtol=1e-5, atol=1e-8
        bad = bad | (np.abs(dm.astype(np.float64) - mask_value)
                     <= 1e-8 + 1e-5 * abs(mask_value))
    shp = dm.shape
    buf = dm.reshape(-1, shp[-1]).astype(np.float64)
    bad2 = bad.reshape(-1, shp[-1]).copy()
    lo, hi = _clip_bounds_rows(buf, ~bad2, sigma, maxiters)
    with np.errstate(invalid='ignore'):
        bad2 |= buf < lo[:, None]
        bad2 |= buf > hi[:, None]
    return np.moveaxis(bad2.reshape(shp), -1, axis)


def _bn_sum32(v):
    """bottleneck's float32 running sum (reduce.c): naive left-to-right"""
    if v.size == 0:
        return np.float32(0)
    return np.cumsum(v, dtype=np.float32)[-1]


def _mean_std(v, accum, ddof=0):
    """(mean, std) of 1-D [v] the way the reference's environment does it.
    accum='f64': float64 accumulators.  accum='bn32' (float32 input only):
    bottleneck.nanmean / nanstd float32 kernels."""
    n = v.size
    if accum == 'bn32' and v.dtype == np.float32:
        mean = _bn_sum32(v) / np.float32(n)
        dev = v - mean
        std = np.sqrt(_bn_sum32(dev * dev) / np.float32(n - ddof))
        return np.float32(mean), np.float32(std)
    v = v.astype(np.float64)
    mean = v.sum() / n
    std = np.sqrt(((v - mean) ** 2).sum() / (n - ddof))
    return mean, std


def sigma_clipped_stats_flat(data, sigma=3.0, maxiters=5, mask_value=None,
                             accum='f64'):
    """sigma_clipped_stats(data, axis=None, cenfunc='mean') -> (mean, std, n):
    SigmaClip._sigmaclip_noaxis (sigma_clipping.py:377-420) followed by
    nanmean / nanstd of the survivors."""
    v = np.asarray(data).ravel()
    if mask_value is not None:
        v = v[~(np.abs(v.astype(np.float64) - mask_value)
                <= 1e-8 + 1e-5 * abs(mask_value))]
    v = v[np.isfinite(v)]
    it = 0
    nchanged = 1
    while nchanged != 0 and it < maxiters:
        it += 1
        if v.size == 0:
            break
        mean, std = _mean_std(v, accum)
        # bounds are formed in float64 from the (float32-valued) mean/std ...
        lo = float(mean) - float(std) * sigma
        hi = float(mean) + float(std) * sigma
        if v.dtype == np.float32 and accum == 'bn32':
            # ... and compared in float32 (numpy 1.x casts the scalar bound to
            # the array's dtype)
            lo, hi = np.float32(lo), np.float32(hi)
            keep = (v >= lo) & (v <= hi)
        else:
            vv = v.astype(np.float64)
            keep = (vv >= lo) & (vv <= hi)
        nchanged = v.size - int(keep.sum())
        v = v[keep]
    if v.size == 0:
        return np.nan, np.nan, 0
    mean, std = _mean_std(v, accum)
    return mean, std, v.size


# --------------------------------------------------------------------------------
# gain / overscan
# --------------------------------------------------------------------------------
def gain_corr(data, gain, ysize_chan, xsize_chan):
    """blackbox.py:7442-7465: data[chan] *= gain[chan] in place.  float32 array
    times python float => float32 multiply by float32(gain)."""
    chan_sec = define_sections(data.shape, ysize_chan, xsize_chan)[0]
    for c in range(16):
        data[chan_sec[c]] *= np.float32(gain[c])


def _polyfit(x, y, deg):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return np.polyfit(x, y, deg)


def hos_mask_ml1(data_hos, data_limit=2000):
    """blackbox.py:6586-6614"""
    mask_hos = data_hos > data_limit
    mask_x = np.sum(mask_hos, axis=0) > 0.5 * mask_hos.shape[0]
    mask_x_open = ndimage.binary_opening(mask_x, structure=np.ones(2))
    mask_hos[:, np.logical_xor(mask_x, mask_x_open)] = False
    return ndimage.binary_dilation(mask_hos, structure=np.ones((3, 3), bool),
                                   iterations=2)


def hos_column_stats(data_hos, mask_hos, accum='f64'):
    """blackbox.py:6649-6678: 2.5-sigma clip down the columns, then per-column
    n, mean, std(ddof=1) -- float32 running sums down the rows in the
    reference (np.nanmean / np.nanstd of a float32 MaskedArray, axis=0)."""
    rej = sigma_clip_axis(data_hos, 0, 2.5, mask=mask_hos)
    ok = ~rej
    n = ok.sum(axis=0)
    if accum == 'bn32':
        f = np.float32
        tot = np.zeros(data_hos.shape[1], f)
        for i in range(data_hos.shape[0]):
            tot = tot + np.where(ok[i], data_hos[i], f(0))
        with np.errstate(invalid='ignore', divide='ignore'):
            mean = tot / n.astype(f)
            dev = np.where(ok, data_hos - mean[None, :], f(0)).astype(f)
            sq = dev * dev
            tot2 = np.zeros_like(tot)
            for i in range(data_hos.shape[0]):
                tot2 = tot2 + sq[i]
            std = np.sqrt(tot2 / (n - 1).astype(f))
    else:
        d = data_hos.astype(np.float64)
        with np.errstate(invalid='ignore', divide='ignore'):
            mean = np.where(ok, d, 0.0).sum(axis=0) / n
            dev = np.where(ok, d - mean[None, :], 0.0)
            std = np.sqrt((dev * dev).sum(axis=0) / (n - 1))
        # the reference keeps these vectors in float32
        mean = mean.astype(np.float32)
        std = std.astype(np.float32)
    return n, mean, std


def hos_fit(n, mean_hos, std_hos, mask_sat_row=None, bg2_chan9=False,
            accum='f64'):
    """blackbox.py:6660-6814: from the per-column clipped (n, mean, std) of the
    horizontal overscan to the float64 vector [oscan] that is subtracted from
    every row of the data section.  mean_hos/std_hos are float32 vectors."""
    ncols = mean_hos.size
    mean_hos = np.asarray(mean_hos, np.float32)
    mask_valid = n > 1
    xcol = np.arange(ncols) + 1
    err_hos = np.zeros(ncols, np.float32)
    with np.errstate(invalid='ignore', divide='ignore'):
        err_hos[mask_valid] = (std_hos[mask_valid].astype(np.float64) /
                               np.sqrt(n[mask_valid])).astype(np.float32)
        weights = np.zeros(ncols, np.float32)
        mask_nonzero = err_hos != 0
        weights[mask_nonzero] = np.float32(1) / err_hos[mask_nonzero]
    if np.all(mask_valid[0:3]):
        weights[0:3] = 0
    idx_switch, overlap = 150, 30
    idx_fit = np.arange(idx_switch + overlap)
    npoints = int(np.sum(mask_valid[idx_fit] & mask_nonzero[idx_fit]))
    m = mask_valid
    y2fit = mean_hos[idx_fit][m[idx_fit]].copy()
    nfit = y2fit.size
    med = [np.median(y2fit[max(k - 1, 3):min(k + 2, nfit)]) for k in range(3, nfit)]
    y2fit[3:] = np.asarray(med, np.float32)
    xs = xcol[idx_fit][m[idx_fit]]
    ws = weights[idx_fit][m[idx_fit]]
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        try:
            splfit = interpolate.UnivariateSpline(xs, y2fit, w=ws, k=2, s=npoints)
        except UserWarning:
            warnings.simplefilter('ignore')
            splfit = interpolate.UnivariateSpline(xs, y2fit, w=ws, k=3,
                                                  s=1.5 * npoints)
    mask_valid_poly = mask_valid.copy()
    mask_valid_poly[0:idx_switch - overlap] = False
    mhp = mean_hos[mask_valid_poly]
    mean, stddev, _ = sigma_clipped_stats_flat(mhp, sigma=5, accum=accum)
    if stddev == 0:
        tmp = np.ones(mhp.size, bool)
    elif accum == 'bn32':
        with np.errstate(invalid='ignore'):
            # float32 array vs python-float-like scalars: float32 arithmetic
            tmp = (np.abs(mhp - np.float32(mean)) / np.float32(stddev)
                   <= np.float32(5))
    else:
        with np.errstate(invalid='ignore'):
            tmp = np.abs(mhp.astype(np.float64) - mean) / stddev <= 5
    mask_valid_poly[mask_valid_poly] = tmp
    err3 = np.float32(3) * err_hos

    def fit_iter(mask_fit, deg):
        fit = None
        for _ in range(3):
            p = _polyfit(xcol[mask_fit], mean_hos[mask_fit], deg)
            fit = np.polyval(p, xcol)
            with np.errstate(invalid='ignore'):
                mask_fit &= (np.abs(fit - mean_hos) <= err3)
        return fit

    if not bg2_chan9:
        oscan = fit_iter(mask_valid_poly, 7)
    else:
        idx_split = 654
        mf = mask_valid_poly.copy()
        mf[idx_split:] = False
        fit1 = fit_iter(mf, 5)
        mf = mask_valid_poly.copy()
        mf[:idx_split] = False
        fit2 = fit_iter(mf, 5)
        oscan = fit1
        oscan[idx_split:] = fit2[idx_split:]
    oscan[0:idx_switch] = splfit(xcol[0:idx_switch])
    oscan[0:3][mask_valid[0:3]] = mean_hos[0:3][mask_valid[0:3]]
    mask_usemean = mask_valid.copy()
    if mask_sat_row is not None:
        mask_usemean &= ~mask_sat_row
    mask_usemean[idx_switch:] = False
    oscan[mask_usemean] = mean_hos[mask_usemean]
    return oscan


def vos_fit(mean_vos_col, nrows, i_chan, poldeg=3):
    """blackbox.py:6497-6556: 5-sigma clean of the row means, exclude the rows
    that overlap the horizontal overscan, polyfit.  Returns (fit float64[dy],
    coefficients low->high, polyfit_ok, mean level)."""
    nrows_chan = mean_vos_col.size
    y_vos = np.arange(nrows_chan)
    polyfit_ok = True
    p = None
    try:
        mean, stddev, _ = sigma_clipped_stats_flat(mean_vos_col, sigma=5)
        if stddev == 0:
            mask_fit = np.ones(nrows_chan, bool)
        else:
            with np.errstate(invalid='ignore'):
                mask_fit = np.abs(mean_vos_col - mean) / stddev <= 5
        if i_chan < 8:
            mask_fit[nrows:] = False
        else:
            mask_fit[:nrows_chan - nrows] = False
        p = np.polyfit(y_vos[mask_fit], mean_vos_col[mask_fit], poldeg)
    except Exception:
        polyfit_ok = False
    if p is None:
        raise RuntimeError('vertical overscan polyfit failed on first channel')
    fit = np.polyval(p, y_vos)
    if not np.all(np.isfinite(fit)):
        polyfit_ok = False
    if polyfit_ok:
        level = np.mean(fit)
    else:
        level = np.nanmedian(mean_vos_col)
        fit = np.full(nrows_chan, level)
    return fit, p[::-1], polyfit_ok, level


class OsCorrError(Exception):
    """what makes the reference's os_corr raise: inside it every warning is an error
    (blackbox.py:6432)"""


def os_corr(data, ysize_chan, xsize_chan, tel='ML1', gain=None, satlevel=None,
            data_limit=2000, accum='f64', ypix_lim=None, header=None):
    """blackbox.py:6407-6879.  [data] float32 raw-shaped array after gain_corr,
    modified in place like the reference does; returns (data_out float32
    without overscans, header dict, aux dict with the 1-D vectors).  [header]: a dict
    to fill (it keeps the keys of the channels done when the function raises, as
    the reference's header does)."""
    chan_sec, data_sec, os_sec_hori, os_sec_vert, data_sec_red = \
        define_sections(data.shape, ysize_chan, xsize_chan)
    ncols, nrows = xsize_chan, ysize_chan
    data_out = np.zeros((nrows * NY, ncols * NX), dtype='float32')
    mean_vos = np.zeros(16)
    std_vos = np.zeros(16)
    header = {} if header is None else header
    aux = dict(vfit=[], oscan=[], dlevel=[], mean_hos=[], std_hos=[], n_hos=[],
               mean_vos_col=[])
    nrows_chan = data[chan_sec[0]].shape[0]
    for c in range(16):
        # ---- vertical overscan ------------------------------------------------
        data_vos = data[os_sec_vert[c]]
        rej = sigma_clip_axis(data_vos, 1, 3.0, mask_value=0)
        if rej.all():
            rej = sigma_clip_axis(data_vos, 1, 3.0)
        d64 = np.where(rej, 0.0, data_vos.astype(np.float64))
        with np.errstate(invalid='ignore', divide='ignore'):
            mean_vos_col = d64.sum(axis=1) / (~rej).sum(axis=1)
        fit_vos_col, coeffs, ok, level = vos_fit(mean_vos_col, nrows, c)
        for k, v in enumerate(coeffs):
            header['BIAS{}A{}'.format(c + 1, k)] = float(v) if np.isfinite(v) else 'None'
        header['VFITOK{}'.format(c + 1)] = bool(ok)
        mean_vos[c] = level
        # float32 array -= float64 column: float64 subtract, round to float32
        sec = data[chan_sec[c]]
        sec[...] = (sec.astype(np.float64) - fit_vos_col.reshape(nrows_chan, 1)
                    ).astype(np.float32)
        # ---- level of the horizontal overscan ---------------------------------
        hos = data[os_sec_hori[c]]
        dlevel, _, _ = sigma_clipped_stats_flat(hos[:, ncols - 300:ncols],
                                                accum=accum)
        hos -= np.float32(dlevel)
        if not np.isfinite(dlevel):
            # an empty window (channels narrower than 300 columns: the slice is taken on
            # the strip incl. the overscan columns) gives a NaN level; the NaN strip
            # overlaps the vertical-overscan section, whose clipped statistics (6572)
            # then warn "Input data contains invalid values" = raise
            raise OsCorrError('channel {}: level of the horizontal overscan is not finite'.format(c + 1))
        _, std_vos[c], _ = sigma_clipped_stats_flat(data[os_sec_vert[c]],
                                                    mask_value=0, accum=accum)
        # ---- horizontal overscan ----------------------------------------------
        data_hos = data[os_sec_hori[c]][:, :ncols]
        mask_sat_row = None
        if tel == 'ML1':
            mask_hos = hos_mask_ml1(data_hos, data_limit)
        else:
            lim = ypix_lim[tel]
            dsec = data[data_sec[c]]
            thr = np.float32(0.9 * (satlevel[c] * gain[c]))
            if c >= 8:
                r1, r2 = slice(0, lim[0]), slice(0, lim[1])
            else:
                r1, r2 = slice(nrows - lim[0], nrows), slice(nrows - lim[1], nrows)
            mask_sat_row = np.sum(dsec[r1, :] >= thr, axis=0) >= 3
            mask_sat_row |= np.sum(dsec[r2, :] >= thr, axis=0) >= 10
            mask_hos = np.zeros(data_hos.shape, bool) | mask_sat_row[None, :]
        n, mean_hos, std_hos = hos_column_stats(data_hos, mask_hos, accum=accum)
        oscan = hos_fit(n, mean_hos, std_hos, mask_sat_row,
                        bg2_chan9=(tel == 'BG2' and c == 8), accum=accum)
        dsec = data[data_sec[c]]
        dsec[...] = (dsec.astype(np.float64) - oscan).astype(np.float32)
        data_out[data_sec_red[c]] = dsec
        aux['vfit'].append(fit_vos_col)
        aux['oscan'].append(oscan)
        aux['dlevel'].append(float(dlevel))
        aux['mean_hos'].append(mean_hos)
        aux['std_hos'].append(std_hos)
        aux['n_hos'].append(n)
        aux['mean_vos_col'].append(mean_vos_col)
    for c in range(16):
        header['BIASM{}'.format(c + 1)] = float(mean_vos[c])
    for c in range(16):
        header['RDN{}'.format(c + 1)] = float(std_vos[c])
    header['BIASMEAN'] = float(np.nanmean(mean_vos))
    header['RDNOISE'] = float(np.nanmean(std_vos))
    return data_out, header, aux


def os_corr_or_zero(data, ysize_chan, xsize_chan, **kw):
    """os_corr inside blackbox_reduce's try / except (blackbox.py:1531-1591): when it raises,
    "adopt an overscan of zero for all channels" = crop the data sections out of the array
    os_corr was modifying in place (channels done before the failure keep their
    correction, the failing one its vertical-overscan subtraction), BIASM{c} = 0,
    RDN{c} = 10, BIASMEAN = 0, RDNOISE = 10, OS-P False.  -> (data_out, header)"""
    header = {}
    try:
        data_out, header, _ = os_corr(data, ysize_chan, xsize_chan, header=header, **kw)
        header['OS-P'] = True
    except Exception:
        _, data_sec, _, _, data_sec_red = define_sections(data.shape, ysize_chan, xsize_chan)
        data_out = np.zeros((ysize_chan * NY, xsize_chan * NX), dtype='float32')
        for c in range(16):
            data_out[data_sec_red[c]] = data[data_sec[c]]
        for c in range(16):
            header['BIASM{}'.format(c + 1)] = 0.0
        for c in range(16):
            header['RDN{}'.format(c + 1)] = 10.0
        header['BIASMEAN'] = 0.0
        header['RDNOISE'] = 10.0
        header['OS-P'] = False
    return data_out, header


# --------------------------------------------------------------------------------
# masks
# --------------------------------------------------------------------------------
def fill_sat_holes(data_mask):
    """blackbox.py:4584-4596"""
    m = ((data_mask & 4) == 4) | ((data_mask & 8) == 8)
    struct = np.ones((3, 3), bool)
    m = ndimage.binary_closing(m, structure=struct)
    m = ndimage.binary_fill_holes(m, structure=struct)
    data_mask[m & (data_mask == 0)] = 8


def mask_init(data, header, bpm, gain, satlevel, ysize_chan, xsize_chan):
    """blackbox.py:4375-4579 (imgtype 'object').  [data] is scrubbed of
    non-finite values in place.  Returns (uint8 mask, header_mask dict)."""
    data_mask = (bpm.copy() if bpm is not None
                 else np.zeros(data.shape, dtype='uint8'))
    header_mask = {}
    mask_infnan = ~np.isfinite(data)
    data[mask_infnan] = 0
    data_mask[mask_infnan & (data_mask == 0)] |= 1
    data_sec_red = define_sections(data.shape, ysize_chan, xsize_chan)[4]
    biaslevel = np.array([header['BIASM{}'.format(c + 1)] for c in range(16)])
    satlevel_chans = np.array(satlevel) * np.array(gain) - biaslevel
    header_mask['SATURATE'] = header['SATURATE'] = float(np.mean(satlevel_chans))
    mask_sat = np.zeros(data.shape, bool)
    for c in range(16):
        key = 'SATLEV{}'.format(c + 1)
        header[key] = header_mask[key] = round(float(satlevel_chans[c]), 1)
        sec = data_sec_red[c]
        # float32 array >= float64 scalar: numpy 1.x compares in float32
        m = data[sec] >= np.float32(satlevel_chans[c])
        mask_sat[sec] = m
        mflip = np.flipud(m)
        for v in range(16):
            if v != c:
                use = m if (c // 8) == (v // 8) else mflip
                data_mask[data_sec_red[v]][use] |= 64
    data_mask[mask_sat] |= 4
    struct = np.ones((3, 3), bool)
    nobj = ndimage.label(mask_sat, structure=struct)[1]
    header_mask['NOBJ-SAT'] = header['NOBJ-SAT'] = int(nobj)
    satcon = ndimage.binary_dilation(mask_sat, structure=struct, iterations=1)
    data_mask[satcon & ~mask_sat] |= 8
    fill_sat_holes(data_mask)
    return data_mask.astype('uint8'), header_mask


def mask_header(data_mask):
    """blackbox.py:4601-4620 -> dict of M-*NUM counts"""
    text = {'bad': 'BP', 'edge': 'EP', 'saturated': 'SP',
            'saturated-connected': 'SCP', 'satellite trail': 'STP',
            'cosmic ray': 'CRP'}
    out = {}
    for k, t in text.items():
        v = MASK_VALUE[k]
        out['M-{}'.format(t)] = True
        out['M-{}VAL'.format(t)] = v
        out['M-{}NUM'.format(t)] = int(np.sum((data_mask & v) == v))
    return out


def edge_fill(data, data_mask, ysize_chan, xsize_chan):
    """blackbox.py:1959-1974: edge pixels <- median of their channel"""
    mask_edge = (data_mask & 32) == 32
    for sec in define_sections(data.shape, ysize_chan, xsize_chan)[4]:
        data[sec][mask_edge[sec]] = np.median(data[sec])


# --------------------------------------------------------------------------------
# crosstalk
# --------------------------------------------------------------------------------
def xtalk_coeffs(rows):
    """rows of (victim, source, correction), 1-based -> coeffs[source, victim]
    (blackbox.py:7196-7198)"""
    coeffs = np.zeros((16, 16))
    for v, s, c in rows:
        coeffs[int(s) - 1, int(v) - 1] = c
    return coeffs


def xtalk_corr(data, coeffs, data_mask, ysize_chan, xsize_chan):
    """blackbox.py:7138-7258, in place on the reduced frame."""
    sec = define_sections(data.shape, ysize_chan, xsize_chan)[4]
    mask_source = (data > 0) & ((data_mask & 1) == 0) & ((data_mask & 2) == 0)
    mask_victim = (data_mask & 32) == 0
    stack = np.stack([data[sec[i]] * mask_source[sec[i]] for i in range(16)], axis=2)
    stack_flip = np.stack([np.flipud(data[sec[i]] * mask_source[sec[i]])
                           for i in range(16)], axis=2)
    corr = np.zeros((16, ysize_chan, xsize_chan))
    s1, s2 = slice(0, 8), slice(8, 16)
    sls = [s1, s2, s1, s2]
    slv = [s1, s1, s2, s2]
    for q in range(4):
        use = stack if q in (0, 3) else stack_flip
        corr[slv[q]] += np.moveaxis(
            np.matmul(use[:, :, sls[q]].astype(np.float64), coeffs[sls[q], slv[q]]),
            2, 0)
    for i in range(16):
        d = data[sec[i]]
        d[...] = (d.astype(np.float64) - corr[i] * mask_victim[sec[i]]
                  ).astype(np.float32)


def nonlin_corr(data, splines, gain, ysize_chan, xsize_chan):
    """blackbox.py:7394-7437, in place on the (overscan-corrected) reduced-shape float32 frame.
    [splines]: 16 callables (scipy UnivariateSpline in the reference's pickle).  Uncorrected
    pixels (counts > 50000) get frac = 1, i.e. are divided by 2 -- as the reference does."""
    sec = define_sections(data.shape, ysize_chan, xsize_chan)[4]
    for i in range(16):
        # float32 array / python float -> float32 (numpy 1.x value-based casting, 7422)
        counts = data[sec[i]] / np.float32(gain[i])
        frac = np.ones(counts.shape)
        m = counts <= 50000
        frac[m] = splines[i](counts[m])
        d = data[sec[i]]
        d[...] = (d.astype(np.float64) / (frac + 1)).astype(np.float32)      # float32 /= float64
    return data


def get_flatstats(data, data_mask, statsec, ysize_chan, xsize_chan, subsize, fraction=None, seed=0):
    """blackbox.py:3661-3820 -> dict of the header values.  [fraction] None: the deterministic
    variant (all valid pixels where the reference draws an unseeded random subsample);
    a number: the reference's estimator with a seeded RandomState (0.2 for the frame, half of
    it for the sub-images), to show the two agree within the sampling error."""
    out = {}
    mask_use = data_mask == 0
    m = mask_use[statsec]
    out['MEDSEC'] = np.nanmedian(data[statsec][m])
    out['STDSEC'] = np.nanstd(data[statsec][m])
    rs = np.random.RandomState(seed)
    if fraction is None:
        sel = data[mask_use]
    else:
        idx = rs.choice(data.size, int(fraction * data.size), replace=False)
        sel = data.ravel()[idx][mask_use.ravel()[idx]]
    out['FLATMED'] = np.nanmedian(sel)
    out['FLATSTD'] = np.nanstd(sel)
    sec = define_sections(data.shape, ysize_chan, xsize_chan)[4]
    for i in range(16):
        out['FLATM%d' % (i + 1)] = np.nanmedian(data[sec[i]])
        out['FLATS%d' % (i + 1)] = np.nanstd(data[sec[i]])
    ns = data.shape[0] // subsize
    dm = np.ma.masked_array(data[:ns * subsize, :ns * subsize], mask=~mask_use[:ns * subsize, :ns * subsize]).reshape(
        ns, subsize, -1, subsize).swapaxes(1, 2).reshape(ns, ns, -1)
    if fraction is None:
        mini_median = np.ma.median(dm, axis=2)
    else:
        idx = rs.choice(dm.shape[2], int(0.5 * fraction * dm.shape[2]), replace=False)
        mini_median = np.ma.median(dm[:, :, idx], axis=2)
    mm = mini_median.reshape(ns, ns, 1)
    dm.mask |= (dm > mm)
    mini_std = np.sqrt(np.ma.sum((dm - mm) ** 2, axis=2) / (np.ma.count(dm, axis=2) - 1))
    from scipy import ndimage
    cn = ndimage.binary_erosion(np.ones(mini_median.shape, dtype=bool))
    mn, mx = np.amin(mini_median[cn]), np.amax(mini_median[cn])
    out['RDIF-MAX'] = np.abs((mx - mn) / (mx + mn))
    nz = mini_median[cn] != 0
    out['RSTD-MAX'] = np.amax(mini_std[cn][nz] / np.abs(mini_median[cn][nz]))
    out['mini_median'] = np.asarray(mini_median)
    out['mini_std'] = np.asarray(mini_std)
    return out


# --------------------------------------------------------------------------------
# master frames
# --------------------------------------------------------------------------------
def master_median(cube, imgtype, medsec=None, bpm=None):
    """blackbox.py:4929-4941, 4984, 5071-5073: [cube] float32 (nfiles, ny, nx), modified in
    place like the reference; returns the master (float32)."""
    if imgtype == 'flat':
        for i in range(cube.shape[0]):
            if medsec[i] != 0:
                cube[i] /= np.float32(medsec[i])
    master = np.median(cube, axis=0)
    if imgtype == 'flat' and bpm is not None:
        master[(bpm == 32) | (master <= 0)] = 1
    return master


def gain_correction_factors(master, ysize_chan, xsize_chan, nrows_v=200, nrows_h=2000, ncols=200):
    """blackbox.py:5085-5161 -> factor_chan[16] (float64).  The float32 master copy is scaled
    in place by float32 scalars (numpy 1.x value-based casting of np.float64/np.float32
    scalars against a float32 array)."""
    sec = define_sections(master.shape, ysize_chan, xsize_chan)[4]
    corr = np.copy(master)
    med = np.zeros(16)
    for i in range(16):
        d = corr[sec[i]]
        med[i] = np.median(d[-nrows_v:, :]) if i < 8 else np.median(d[0:nrows_v, :])
        d /= np.float32(med[i])
    factor = 1. / med
    dy, dx = ysize_chan, xsize_chan
    for i in range(1, 8):
        y_index, x_index = dy, i * dx
        s1 = corr[y_index - nrows_h:y_index + nrows_h, x_index - ncols:x_index]
        s2 = corr[y_index - nrows_h:y_index + nrows_h, x_index:x_index + ncols]
        ratio = np.median(s1) / np.nanmedian(s2)
        corr[sec[i]] *= np.float32(ratio)
        corr[sec[i + 8]] *= np.float32(ratio)
        factor[i] *= ratio
        factor[i + 8] *= ratio
    factor /= np.mean(factor)
    return factor


def sigma_clipped_stats_median(x, sigma=3.0, maxiters=5, mask_value=0):
    """astropy.stats.sigma_clipped_stats(x, mask_value=0) (astropy 4.3 defaults: centre
    median, spread std ddof 0, closed interval) -> (mean, median, std, n) of the survivors"""
    v = np.asarray(x).ravel()
    v = v[np.isfinite(v) & (v != mask_value)]
    for _ in range(maxiters):
        n = v.size
        if n == 0:
            break
        c, s = np.median(v), np.std(v.astype(np.float64))
        v = v[(v >= c - sigma * s) & (v <= c + sigma * s)]
        if v.size == n:
            break
    return np.mean(v.astype(np.float64)), np.median(v), np.std(v.astype(np.float64)), v.size
