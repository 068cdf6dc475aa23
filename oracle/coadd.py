"""oracle/coadd.py -- CPU restatement of the reference co-add path.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and tools/stage_bench.py as the checker), never by
the product path.

Two kinds of code live here:

* in-reference arithmetic, restated operation by operation (numpy float32 semantics):
    prep_weights, prep_data   buildref.py:2602-2624, 2709-2733 (prep_inputimages)
    scale_chan_zps            buildref.py:3019-3047
  Their parity is pinned by the reference's own lines: the same numpy expressions on the same
  arrays.

* [EXT] SWarp 2.38 (called through subprocess at buildref.py:1727-1833 with
  -RESAMPLING_TYPE LANCZOS3 and -COMBINE_TYPE WEIGHTED / CLIPPED / MEDIAN ...).  SWarp is a
  third-party C program that is not under /root/reference and not installed here: the
  resampling and combination below follow its published algorithm (Bertin et al. 2002; the
  SWarp user's guide 2.21 section 6; CLIPPED: Gruen, Seitz & Bernstein 2014) as remembered.
  PARITY UNPINNED for lanczos3_resample and combine.

Conventions: images are C-order float32 [ny][nx]; pixel coordinates are 0-based, pixel (j, i)
covers [i-0.5, i+0.5); xin/yin give for every output pixel the input position of its centre.
"""
import numpy as np

F = np.float32
BIG = 1e30


# ---------------------------------------------------------------------------------------
# in-reference arithmetic (buildref.py prep_inputimages)
# ---------------------------------------------------------------------------------------
def prep_weights(bkg_std, mask, discard_bits, nimages=2):
    """buildref.py:2602-2624: weights = 1/bkg_std**2 where bkg_std != 0, then 0 where the mask
    holds one of the discarded types (only when more than one image is combined)"""
    w = np.zeros_like(bkg_std, dtype='float32')
    nz = np.nonzero(bkg_std)
    w[nz] = 1 / (bkg_std[nz]) ** 2
    if nimages > 1:
        w[(mask & np.uint8(discard_bits)) != 0] = 0
    return w


def prep_data(data, bkg, mask, edge_value=32):
    """buildref.py:2709-2733: background subtraction, then pixels whose mask EQUALS the edge
    value are set to zero (an equality, not a bit test)"""
    out = np.array(data, dtype='float32', copy=True)
    if bkg is not None:
        out -= bkg
    out[mask == edge_value] = 0
    return out


def scale_chan_zps(data, zp, zp_chan, ysz, xsz):
    """buildref.py:3019-3047: data[channel c] *= 10**(0.4*(zp - zp_chan[c])) (python float factor
    on a float32 array; channels without a zeropoint keep factor 1: zp_chan[c] = None)"""
    out = np.array(data, dtype='float32', copy=True)
    for c in range(16):
        if zp_chan[c] is None:
            continue
        iy, ix = c // 8, c % 8
        out[iy * ysz:(iy + 1) * ysz, ix * xsz:(ix + 1) * xsz] *= 10 ** (0.4 * (zp - zp_chan[c]))
    return out


# ---------------------------------------------------------------------------------------
# [EXT] SWarp: LANCZOS3 resampling
# ---------------------------------------------------------------------------------------
def lanczos3_taps(frac):
    """normalised kernel values at the 6 taps ix-2 .. ix+3 for a position ix + frac
    (0 <= frac < 1): k(t) = sinc(t) sinc(t/3) for |t| < 3, then divided by their sum
    (SWarp normalises every interpolation kernel to unit sum).  float64 -> float32."""
    frac = np.asarray(frac, np.float64)
    t = frac[..., None] - np.arange(-2, 4)
    k = np.sinc(t) * np.sinc(t / 3.0)
    k[np.abs(t) >= 3.0] = 0.0
    k /= k.sum(axis=-1, keepdims=True)
    return k.astype(F)


def lanczos3_resample(img, wimg, xin, yin, fscale=1.0):
    """-> (out float32, wout float32).  For every output pixel with input position (xin, yin):
    the 6x6 taps around floor(x), floor(y); a footprint that leaves the input image gives
    (0, weight 0).  Data: out = fscale * sum ky kx f (float32 products accumulated in float32,
    rows outermost).  Weights travel as variances (1/w, BIG where w == 0), are interpolated
    with the same kernel and scaled by fscale**2; a footprint holding a zero-weight pixel or a
    non-positive interpolated variance gives weight 0."""
    ny, nx = img.shape
    xin = np.asarray(xin, np.float64); yin = np.asarray(yin, np.float64)
    ix = np.floor(xin).astype(np.int64); iy = np.floor(yin).astype(np.int64)
    kx = lanczos3_taps(xin - ix); ky = lanczos3_taps(yin - iy)
    inside = (ix - 2 >= 0) & (ix + 3 < nx) & (iy - 2 >= 0) & (iy + 3 < ny)
    ixc = np.where(inside, ix, 2); iyc = np.where(inside, iy, 2)
    acc = np.zeros(xin.shape, F)
    vacc = np.zeros(xin.shape, F)
    bad = np.zeros(xin.shape, bool)
    var = np.where(wimg > 0, F(1) / np.where(wimg > 0, wimg, F(1)), F(BIG)).astype(F)
    for j in range(6):
        row = np.zeros(xin.shape, F); vrow = np.zeros(xin.shape, F)
        for i in range(6):
            f = img[iyc + j - 2, ixc + i - 2]
            v = var[iyc + j - 2, ixc + i - 2]
            row = row + kx[..., i] * f
            vrow = vrow + kx[..., i] * v
            bad |= wimg[iyc + j - 2, ixc + i - 2] <= 0
        acc = acc + ky[..., j] * row
        vacc = vacc + ky[..., j] * vrow
    fs = F(fscale)
    out = (fs * acc).astype(F)
    vout = (fs * fs) * vacc
    ok = inside & ~bad & (vout > 0)
    wout = np.where(ok, F(1) / np.where(ok, vout, F(1)), F(0)).astype(F)
    out = np.where(inside, out, F(0)).astype(F)
    return out, wout


def coarse_grid(xfun, out_ny, out_nx, step):
    """exact input coordinates on the nodes of a coarse grid over the output frame (what SWarp
    does when PROJECTION_ERR > 0: the projection is evaluated on a lattice and interpolated).
    xfun(yout, xout) -> (xin, yin).  Nodes at 0, step, 2 step, ... up to and past the last pixel.
    -> grid float64 [gny][gnx][2] (x, y)"""
    gy = np.arange(0, out_ny + step, step, dtype=np.float64)
    gx = np.arange(0, out_nx + step, step, dtype=np.float64)
    yy, xx = np.meshgrid(gy, gx, indexing='ij')
    xin, yin = xfun(yy, xx)
    return np.stack([xin, yin], axis=-1).astype(np.float64)


def grid_positions(grid, out_ny, out_nx, step):
    """bilinear interpolation (float64) of the coarse grid at every output pixel"""
    j = np.arange(out_ny)[:, None]; i = np.arange(out_nx)[None, :]
    gj, gi = j // step, i // step
    fy = (j - gj * step) / float(step); fx = (i - gi * step) / float(step)
    res = []
    for k in range(2):
        g = grid[..., k]
        a = g[gj, gi] + (g[gj, gi + 1] - g[gj, gi]) * fx
        b = g[gj + 1, gi] + (g[gj + 1, gi + 1] - g[gj + 1, gi]) * fx
        res.append(a + (b - a) * fy)
    return res[0], res[1]


# ---------------------------------------------------------------------------------------
# [EXT] SWarp: combination of the resampled images
# ---------------------------------------------------------------------------------------
def combine(cube, wcube, combine_type='weighted', clip_sigma=4.0, clip_ampfrac=0.3):
    """cube, wcube float32 [n][ny][nx]; only pixels with weight > 0 take part.
    -> (out, wout, nclip[n])

    weighted : sum(w f) / sum(w),            weight sum(w)
    average  : mean(f),                      weight m**2 / sum(1/w)
    median   : np.median(f) (mean of the two middle values for an even count),
               weight (2/pi) m**2 / sum(1/w)  (asymptotic efficiency of the median)
    min/max  : extreme value,                weight of that pixel
    sum      : sum(f),                       weight 1 / sum(1/w)
    clipped  : reference value = median of the valid values; a value is dropped when
               |f - med| > clip_sigma * sqrt(1/w) + clip_ampfrac * |med| (float32); then `weighted`
               over the rest (never drops everything: the median itself survives for odd m;
               if nothing survives the unclipped weighted mean is used)
    float64 accumulation over the images in order, results rounded to float32."""
    n = cube.shape[0]
    c = cube.astype(np.float64); w = wcube.astype(np.float64)
    valid = w > 0
    m = valid.sum(axis=0)
    nclip = np.zeros(n, np.int64)
    t = combine_type.lower()
    with np.errstate(divide='ignore', invalid='ignore'):
        sw = np.where(valid, w, 0.0).sum(axis=0)
        swf = np.where(valid, w * c, 0.0).sum(axis=0)
        sinv = np.where(valid, 1.0 / np.where(valid, w, 1.0), 0.0).sum(axis=0)
        if t == 'weighted':
            out = np.where(m > 0, swf / sw, 0.0); wout = sw
        elif t == 'average':
            out = np.where(m > 0, np.where(valid, c, 0.0).sum(axis=0) / m, 0.0)
            wout = np.where(m > 0, m * m / sinv, 0.0)
        elif t == 'sum':
            out = np.where(valid, c, 0.0).sum(axis=0)
            wout = np.where(m > 0, 1.0 / sinv, 0.0)
        elif t in ('min', 'max'):
            fill = np.inf if t == 'min' else -np.inf
            cc = np.where(valid, c, fill)
            idx = cc.argmin(axis=0) if t == 'min' else cc.argmax(axis=0)
            out = np.where(m > 0, np.take_along_axis(cc, idx[None], 0)[0], 0.0)
            wout = np.where(m > 0, np.take_along_axis(w, idx[None], 0)[0], 0.0)
        elif t in ('median', 'clipped'):
            cs = np.sort(np.where(valid, c, np.inf), axis=0)
            lo = np.take_along_axis(cs, np.maximum((m - 1) // 2, 0)[None], 0)[0]
            hi = np.take_along_axis(cs, np.maximum(m // 2, 0)[None], 0)[0]
            med = np.where(m > 0, 0.5 * (lo + hi), 0.0)
            if t == 'median':
                out = med
                wout = np.where(m > 0, (2.0 / np.pi) * m * m / sinv, 0.0)
            else:
                # the clip test in float32 (SWarp's pixel type), operations in this order
                med32 = med.astype(F)
                sig = np.sqrt(F(1) / np.where(valid, wcube, F(1)).astype(F))
                thr = F(clip_sigma) * sig + (F(clip_ampfrac) * np.abs(med32))[None]
                drop = valid & (np.abs(cube.astype(F) - med32[None]) > thr)
                keep = valid & ~drop
                nk = keep.sum(axis=0)
                sw2 = np.where(keep, w, 0.0).sum(axis=0)
                swf2 = np.where(keep, w * c, 0.0).sum(axis=0)
                out = np.where(nk > 0, swf2 / np.where(nk > 0, sw2, 1.0), np.where(m > 0, swf / sw, 0.0))
                wout = np.where(nk > 0, sw2, sw)
                nclip = (drop & (nk > 0)[None]).sum(axis=(1, 2))
        else:
            raise ValueError('combine_type ' + combine_type)
    out = np.where(m > 0, out, 0.0)
    return out.astype(F), np.where(m > 0, wout, 0.0).astype(F), nclip


def coadd(images, weights, positions, fscales, combine_type='weighted', **kw):
    """resample every image onto the output frame and combine: positions[k] = (xin, yin)"""
    cube, wcube = [], []
    for img, w, (xin, yin), fs in zip(images, weights, positions, fscales):
        o, wo = lanczos3_resample(img, w, xin, yin, fs)
        cube.append(o); wcube.append(wo)
    return combine(np.stack(cube), np.stack(wcube), combine_type, **kw)


# ---------------------------------------------------------------------------------------
# in-reference: clipped pixels of the first CLIPPED pass -> masks in the input frames
# (buildref.py clipped2mask_loop 3686-3783, pass_filters 3784-3873)
# ---------------------------------------------------------------------------------------
def clip_nsigma(cube, wcube, clip_sigma=4.0, clip_ampfrac=0.3):
    """what the clip log holds per dropped pixel [EXT SWarp -CLIP_WRITELOG: image, x, y, deviation in
    sigma]: (f - median) / sqrt(1/w) in float32 where `combine(..., 'clipped')` drops a pixel, else 0"""
    valid = wcube > 0
    m = valid.sum(axis=0)
    cs = np.sort(np.where(valid, cube.astype(np.float64), np.inf), axis=0)
    lo = np.take_along_axis(cs, np.maximum((m - 1) // 2, 0)[None], 0)[0]
    hi = np.take_along_axis(cs, np.maximum(m // 2, 0)[None], 0)[0]
    med32 = np.where(m > 0, 0.5 * (lo + hi), 0.0).astype(F)
    with np.errstate(divide='ignore', invalid='ignore'):
        sig = np.sqrt(F(1) / np.where(valid, wcube, F(1)).astype(F))
        thr = F(clip_sigma) * sig + (F(clip_ampfrac) * np.abs(med32))[None]
        dev = cube.astype(F) - med32[None]
        drop = valid & (np.abs(dev) > thr)
        nk = (valid & ~drop).sum(axis=0)
        drop &= (nk > 0)[None]
        return np.where(drop, dev / sig, F(0)).astype(F), drop


def pass_filters(x, y, nsigma, fsize, fsigma, fmax, mask_shape):
    """buildref.py:3784-3873 written out (x, y 1-based integer pixel positions of the clipped pixels
    of one image, nsigma their deviations) -> bool mask"""
    x = np.asarray(x, np.int64); y = np.asarray(y, np.int64); nsigma = np.asarray(nsigma, np.float32)
    mask_im = np.zeros(mask_shape, dtype=bool)
    for nf in range(len(fsize)):
        sel = np.abs(nsigma) > fsigma[nf]
        xs, ys, ns = x[sel], y[sel], nsigma[sel]
        keep = ~mask_im[ys - 1, xs - 1]
        xs, ys, ns = xs[keep], ys[keep], ns[keep]
        x_index, y_index = xs - 1, ys - 1
        if fsize[nf] == 1:
            mask_im[y_index, x_index] = True
        else:
            ysize, xsize = mask_shape
            count_im = np.zeros((2, ysize, xsize), dtype='uint16')
            count_index = (ns > 0).astype(np.int64)
            for it in range(xs.size):
                i0, j0 = x_index[it], y_index[it]
                i1, j1 = min(i0 + fsize[nf], xsize), min(j0 + fsize[nf], ysize)
                count_im[count_index[it], j0:j1, i0:i1] += 1
            mask_count = (count_im[0] >= fmax[nf]) | (count_im[1] >= fmax[nf])
            for y1, x1 in zip(*np.nonzero(mask_count)):
                i1, j1 = x1 + 1, y1 + 1
                i0, j0 = max(i1 - fsize[nf], 0), max(j1 - fsize[nf], 0)
                mask_im[j0:j1, i0:i1] = True
    return mask_im


def clipped2mask(clip_out, nsig_out, xin, yin, in_shape, data_mask, weights, nsigma_clip, fwhm,
                 sat_bits=12, fsize=(5, 1), fmax=(4, 1)):
    """clipped2mask_loop (buildref.py:3686-3783) for one input image: the clipped pixels of the
    output frame (clip_out bool, nsig_out float32) are carried to the input frame -- (xin, yin) =
    input position of every output pixel, 0-based; the reference goes through sky coordinates, the
    rounding `(x_im + 0.5).astype(uint16)` on 1-based positions is the same -- filtered
    (fsigma = [nsigma_clip, 4]), those within 5 FWHM of a saturated / saturated-connected pixel are
    released, and the weights of the rest are set to zero.  -> (mask bool, weights)"""
    ysize, xsize = in_shape
    jj, ii = np.nonzero(clip_out)
    ns = nsig_out[jj, ii]
    keep0 = np.abs(ns) > min(nsigma_clip, 4)
    jj, ii, ns = jj[keep0], ii[keep0], ns[keep0]
    x1 = xin[jj, ii] + 1.0; y1 = yin[jj, ii] + 1.0
    ok = np.isfinite(x1) & np.isfinite(y1) & (x1 + 0.5 >= 0) & (y1 + 0.5 >= 0) & (x1 < 60000) & (y1 < 60000)
    xi = np.zeros(x1.shape, np.int64); yi = np.zeros(x1.shape, np.int64)
    xi[ok] = (x1[ok] + 0.5).astype(np.int64); yi[ok] = (y1[ok] + 0.5).astype(np.int64)
    keep = ok & (xi >= 1) & (xi <= xsize) & (yi >= 1) & (yi <= ysize)
    mask_im = pass_filters(xi[keep], yi[keep], ns[keep], list(fsize), [nsigma_clip, 4], list(fmax), in_shape)
    mask_sat = (data_mask & np.uint8(sat_bits)) != 0
    y_sat, x_sat = np.nonzero(mask_sat)
    y_im, x_im = np.nonzero(mask_im)
    dist2_limit = (5 * fwhm) ** 2
    for i in range(y_sat.size):
        near = (x_im - x_sat[i]) ** 2 + (y_im - y_sat[i]) ** 2 <= dist2_limit
        if near.any():
            mask_im[y_im[near], x_im[near]] = False
    w = weights.copy()
    w[mask_im] = 0
    return mask_im, w
