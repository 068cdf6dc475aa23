"""TEST INFRASTRUCTURE -- CPU restatement of the ZOGY numerical core that BlackBOX calls.

PARITY UNPINNED: `zogy.optimal_subtraction` (pmvreeswijk/ZOGY, un-pinned, not even listed
in the reference's pyproject.toml) is absent from /root/reference and from this container;
the reference holds no test or golden vector for it.  Anchors: the call sites
blackbox.py:2350-2354 / 2460-2465, the helper usage in buildref.py (box reshape + nanmedian
2398-2405, mini2back(data_mini, data_shape, order_interp=3, bkg_boxsize, interp_Xchan)
2480-2495), the product/keyword lists (set_blackbox.py:157-164, blackbox.py:3058-3066,
3168-3185), SURVEY.md Appendix A.3-A.5 and Zackay, Ofek & Gal-Yam 2016 (ApJ 830, 27).

Conventions fixed here (and followed by the HIP kernels):
* get_back: per bkg_boxsize box, astropy-style sigma clipping of the unmasked, non-zero
  pixels (centre = median, std about the mean, 3 sigma, <= 5 iterations) -> median and std
  of the survivors; boxes with < half of their pixels usable are NaN, then filled with the
  nan-median of their 3x3 neighbours (repeated), then a 3x3 median filter (edge replicated).
* mini2back: scipy.ndimage.zoom(mini, boxsize, order=3, mode='nearest'); per channel when
  interp_Xchan is False.
* run_zogy: the published D / S / S_corr / F_psf expressions, float32 / complex64.
"""
import numpy as np
from scipy import ndimage

F = np.float32


# --------------------------------------------------------------------------------
# background mesh
# --------------------------------------------------------------------------------
def box_stats(values, nsigma=3.0, maxiters=5):
    """values: 1-D float64 of the usable pixels of one box -> (median, std) float64"""
    v = np.sort(values)
    for _ in range(maxiters):
        if v.size == 0:
            break
        med = np.median(v)
        mean = v.sum() / v.size
        std = np.sqrt(((mean - v) ** 2).sum() / v.size)
        keep = (v >= med - nsigma * std) & (v <= med + nsigma * std)
        if keep.all():
            break
        v = v[keep]
    if v.size == 0:
        return np.nan, np.nan
    mean = v.sum() / v.size
    return np.median(v), np.sqrt(((v - mean) ** 2).sum() / v.size)


def get_back_mini(data, mask, objmask=None, box=60, limfrac=0.5):
    """-> (mini_median, mini_std) float32 arrays of shape (ny/box, nx/box); NaN = too few pixels"""
    ny, nx = data.shape
    nby, nbx = ny // box, nx // box
    reject = (mask != 0)
    if objmask is not None:
        reject = reject | (objmask != 0)
    reject = reject | (data == 0)                       # mask_value = 0
    med = np.full((nby, nbx), np.nan)
    std = np.full((nby, nbx), np.nan)
    for by in range(nby):
        for bx in range(nbx):
            sl = (slice(by * box, (by + 1) * box), slice(bx * box, (bx + 1) * box))
            v = data[sl][~reject[sl]].astype(np.float64)
            if v.size >= limfrac * box * box:
                med[by, bx], std[by, bx] = box_stats(v)
    return med.astype(F), std.astype(F)


def fill_filter_mini(mini, filtsize=3):
    """NaN boxes <- nan-median of the 3x3 neighbourhood (values of the previous sweep),
    repeated until none is left; then a 3x3 median filter with replicated edges."""
    m = mini.astype(F).copy()
    for _ in range(m.size):
        bad = np.isnan(m)
        if not bad.any() or bad.all():
            break
        prev = np.pad(m, 1, mode='constant', constant_values=np.nan)
        new = m.copy()
        for y, x in zip(*np.nonzero(bad)):
            w = prev[y:y + 3, x:x + 3].ravel()
            w = w[~np.isnan(w)]
            if w.size:
                new[y, x] = F(np.median(w.astype(np.float64)))
        m = new
    return ndimage.median_filter(m, size=filtsize, mode='nearest')


def mini2back(mini, shape, box, channels=None):
    """bicubic-spline zoom of the mini image to the full frame.  channels=(ny_chan, nx_chan)
    in boxes -> each channel block is zoomed on its own (interp_Xchan=False)."""
    if channels is None:
        out = ndimage.zoom(mini, box, order=3, mode='nearest')
    else:
        cy, cx = channels
        out = np.empty((mini.shape[0] * box, mini.shape[1] * box), F)
        for y in range(0, mini.shape[0], cy):
            for x in range(0, mini.shape[1], cx):
                out[y * box:(y + cy) * box, x * box:(x + cx) * box] = ndimage.zoom(
                    mini[y:y + cy, x:x + cx], box, order=3, mode='nearest')
    assert out.shape == tuple(shape)
    return out.astype(F)


# --------------------------------------------------------------------------------
# ZOGY per sub-image
# --------------------------------------------------------------------------------
def run_zogy(N, R, Pn, Pr, sn, sr, fn, fr, Vn, Vr, dx, dy):
    """All images float32 (L, L); Pn, Pr centred on pixel [0,0] (ifftshift-ed), unit sum.
    -> D, S, Scorr, Fpsf, Fpsferr (float32)"""
    N, R, Pn, Pr, Vn, Vr = (np.asarray(a, F) for a in (N, R, Pn, Pr, Vn, Vr))
    f2, if2 = np.fft.fft2, np.fft.ifft2
    Nh, Rh, Pnh, Prh = f2(N), f2(R), f2(Pn), f2(Pr)
    sn, sr, fn, fr, dx, dy = (F(a) for a in (sn, sr, fn, fr, dx, dy))
    Pn2 = (Pnh.real ** 2 + Pnh.imag ** 2).astype(F)
    Pr2 = (Prh.real ** 2 + Prh.imag ** 2).astype(F)
    sn2, sr2, fn2, fr2 = sn * sn, sr * sr, fn * fn, fr * fr
    fD = fr * fn / np.sqrt(sn2 * fr2 + sr2 * fn2)
    den = (sn2 * fr2) * Pr2 + (sr2 * fn2) * Pn2
    sden = np.sqrt(den)
    Dh = (fr * (Prh * Nh) - fn * (Pnh * Rh)) / sden
    D = (if2(Dh).real / fD).astype(F)
    PDh = (fr * fn / fD) * (Prh * Pnh) / sden
    Sh = fD * Dh * np.conj(PDh)
    S = if2(Sh).real.astype(F)
    krh = fr * fn2 * np.conj(Prh) * Pn2 / den
    knh = fn * fr2 * np.conj(Pnh) * Pr2 / den
    kr = if2(krh).real.astype(F)
    kn = if2(knh).real.astype(F)
    VSr = if2(f2(Vr) * f2(kr * kr)).real.astype(F)
    VSn = if2(f2(Vn) * f2(kn * kn)).real.astype(F)
    Sn = if2(knh * Nh).real.astype(F)
    Sr = if2(krh * Rh).real.astype(F)
    dSndy = Sn - np.roll(Sn, 1, axis=0)
    dSndx = Sn - np.roll(Sn, 1, axis=1)
    dSrdy = Sr - np.roll(Sr, 1, axis=0)
    dSrdx = Sr - np.roll(Sr, 1, axis=1)
    Vast = dx * dx * (dSndx ** 2 + dSrdx ** 2) + dy * dy * (dSndy ** 2 + dSrdy ** 2)
    VS = VSr + VSn
    with np.errstate(invalid='ignore', divide='ignore'):
        Scorr = (S / np.sqrt(VS + Vast)).astype(F)
    FS = F(np.sum((fn2 * Pn2 * fr2 * Pr2 / den).astype(np.float64)) / N.size)
    Fpsf = (S / FS).astype(F)
    Fpsferr = (np.sqrt(np.maximum(VS, 0)) / FS).astype(F)
    return D, S, Scorr, Fpsf, Fpsferr


def subimage_grid(ny, nx, size, border):
    """cut-outs of (size + 2*border)^2 around a regular grid of size^2 tiles, clamped at the
    frame edge by shifting inwards is NOT done: pixels outside the frame are zero-padded.
    -> list of (y0, x0) of the padded cut-out origin"""
    return [(y - border, x - border) for y in range(0, ny, size) for x in range(0, nx, size)]


def cut_subimages(img, size, border):
    ny, nx = img.shape
    L = size + 2 * border
    pad = np.pad(img, border, mode='constant')
    return np.stack([pad[y + border:y + border + L, x + border:x + border + L]
                     for (y, x) in subimage_grid(ny, nx, size, border)]).astype(F)


def stitch_subimages(subs, ny, nx, size, border):
    out = np.empty((ny, nx), F)
    k = 0
    for y in range(0, ny, size):
        for x in range(0, nx, size):
            out[y:y + size, x:x + size] = subs[k][border:border + size, border:border + size]
            k += 1
    return out


# --------------------------------------------------------------------------------
# PSF photometry
# --------------------------------------------------------------------------------
def psf_optflux(D, V, psfs, ys, xs):
    """optimal (PSF-weighted) flux at integer positions: stamps of the unit-sum PSF model
    [nsrc, S, S] centred on (ys, xs): flux = sum(P*D/V) / sum(P^2/V), err = 1/sqrt(sum(P^2/V));
    pixels outside the frame or with V <= 0 are skipped.  float64 accumulators -> float32."""
    nsrc, S, _ = psfs.shape
    h = S // 2
    flux = np.zeros(nsrc, F)
    err = np.zeros(nsrc, F)
    ny, nx = D.shape
    for k in range(nsrc):
        num = den = 0.0
        for j in range(S):
            for i in range(S):
                y, x = ys[k] + j - h, xs[k] + i - h
                if 0 <= y < ny and 0 <= x < nx and V[y, x] > 0:
                    p = np.float64(psfs[k, j, i])
                    num += p * np.float64(D[y, x]) / np.float64(V[y, x])
                    den += p * p / np.float64(V[y, x])
        flux[k] = F(num / den) if den > 0 else F(0)
        err[k] = F(1.0 / np.sqrt(den)) if den > 0 else F(0)
    return flux, err


def psf_optflux_vec(D, V, psfs, ys, xs):
    """psf_optflux for many sources: the same sums (float64, pixels outside the frame or with V <= 0
    skipped) formed with array operations per source; psfs [nsrc, S, S] or one stamp [S, S] for all"""
    D = np.asarray(D); V = np.asarray(V)
    one = np.ndim(psfs) == 2
    S = psfs.shape[-1]
    h = S // 2
    ny, nx = D.shape
    nsrc = len(ys)
    flux = np.zeros(nsrc, F); err = np.zeros(nsrc, F)
    for k in range(nsrc):
        y0, x0 = int(ys[k]) - h, int(xs[k]) - h
        ja, jb, ia, ib = max(0, -y0), min(S, ny - y0), max(0, -x0), min(S, nx - x0)
        if ja >= jb or ia >= ib:
            continue
        p = (psfs if one else psfs[k])[ja:jb, ia:ib].astype(np.float64)
        d = D[y0 + ja:y0 + jb, x0 + ia:x0 + ib].astype(np.float64)
        v = V[y0 + ja:y0 + jb, x0 + ia:x0 + ib].astype(np.float64)
        ok = v > 0
        num = (p[ok] * d[ok] / v[ok]).sum()
        den = (p[ok] * p[ok] / v[ok]).sum()
        if den > 0:
            flux[k], err[k] = F(num / den), F(1.0 / np.sqrt(den))
    return flux, err


def find_transients_fast(Scorr, nsigma=6.0, regions=False):
    """find_transients for big frames: the same regions and peaks through ndimage.find_objects (the C order of a
    region's pixels inside its bounding box is their C order in the frame: same pixel on ties).
    regions=True -> (label image, [(y, x, peak, (y0, y1, x0, x1)) per label 1..n]): the bounding boxes as well"""
    a = np.abs(Scorr)
    lab, n = ndimage.label(a >= nsigma, structure=np.ones((3, 3), bool))
    out = []
    for k, sl in enumerate(ndimage.find_objects(lab), start=1):
        sub = np.where(lab[sl] == k, a[sl], -1.0)
        j, i = np.unravel_index(np.argmax(sub), sub.shape)
        y, x = sl[0].start + int(j), sl[1].start + int(i)
        out.append((y, x, float(Scorr[y, x]), (sl[0].start, sl[0].stop, sl[1].start, sl[1].stop)))
    if regions:
        return lab, out
    return sorted(t[:3] for t in out)


def find_transients(Scorr, nsigma=6.0):
    """connected regions (8-conn) of |Scorr| >= nsigma -> list of (y, x, peak) of the pixel
    with the largest |Scorr| per region (first in C order on ties), sorted by (y, x)"""
    sig = np.abs(Scorr) >= nsigma
    lab, n = ndimage.label(sig, structure=np.ones((3, 3), bool))
    out = []
    a = np.abs(Scorr)
    for k in range(1, n + 1):
        ys, xs = np.nonzero(lab == k)
        i = np.argmax(a[ys, xs])
        out.append((int(ys[i]), int(xs[i]), float(Scorr[ys[i], xs[i]])))
    return sorted(out)


def psf_model(terms, basis):
    """stamp[s][p] = sum_k terms[s][k] * basis[k][p] as a k-ordered float32 fma chain (what
    the f32 MFMA computes).  fma(a, b, c) in float32 = round32(a*b + c): a*b is exact in
    float64, the float64 sum is rounded once more to float32 -- identical to a true fma except
    in rare double-rounding cases (tests allow 1 ulp there)."""
    terms = np.asarray(terms, np.float32); basis = np.asarray(basis, np.float32)
    acc = np.zeros((terms.shape[0], basis.shape[1]), np.float32)
    for k in range(terms.shape[1]):
        acc = (terms[:, k:k + 1].astype(np.float64) * basis[k][None, :].astype(np.float64) + acc.astype(np.float64)
               ).astype(np.float32)
    return acc


def get_psf_ima(basis, x, y, psf_samp, polzero, polscal, poldeg):
    """zogy.get_psf_ima as called at buildref.py:3357-3366 [EXT: zogy is absent; restated from its call signature and the
    PSFEx file format]: the PSFEx model (basis planes tabulated every [psf_samp] image pixels) evaluated at image
    position (x, y) -> the polynomial combination of the planes on the model's own grid, then resampled to image
    pixels: psf_size = ceil(S_config * psf_samp) made odd, scipy.ndimage.zoom by psf_size / S_config (order 2,
    mode 'nearest'), normalised to unit sum.  -> float32 [psf_size, psf_size]"""
    basis = np.asarray(basis, np.float64)
    xn = (x - polzero[0]) / polscal[0]
    yn = (y - polzero[1]) / polscal[1]
    ima = np.zeros(basis.shape[1:])
    k = 0
    for j in range(poldeg + 1):
        for i in range(poldeg + 1 - j):
            ima += basis[k] * (xn ** i) * (yn ** j)
            k += 1
    s_cfg = basis.shape[1]
    size = int(np.ceil(s_cfg * psf_samp))
    size += 1 - size % 2
    out = ndimage.zoom(ima, size / s_cfg, order=2, mode='nearest') if size != s_cfg else ima
    assert out.shape == (size, size)
    return (out / out.sum()).astype(F)
