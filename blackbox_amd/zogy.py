"""Numerical core of zogy.optimal_subtraction on the GPU: background mesh, sub-image ZOGY
(rocFFT), PSF photometry.  Call sites in the reference: blackbox.py:2350-2354 / 2460-2465;
helper signatures seen in buildref.py:2398-2405, 2480-2495, 3357-3366.

[EXT] zogy itself is not part of /root/reference: the conventions are those of
oracle/zogy_core.py (parity unpinned).  Astrometry, PSFEx,
SExtractor and the real-bogus CNN stay out of scope: PSF images and a WCS-aligned
reference frame are inputs.
"""
import ctypes as C

import numpy as np
import torch
from scipy import ndimage

from . import settings
from ._lib import lib, check, fetch, push, BBXError as _lib_BBXError, SplineImage as _SplineImage
from .catalogs import format_cat, transient_table         # noqa: F401  (zogy.format_cat)

BBX_ERR_OVERFLOW, BBX_ERR_PSFWIN = -4, -6          # include/bbx.h
BBX_OPT_ZOGY_KWIN_OFF = 4
NPAD = 12          # scipy.ndimage.zoom pads 'nearest' inputs by 12 samples before prefiltering


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# ---- background mesh --------------------------------------------------------------------
def get_back(ctx, data, data_mask, objmask=None, bkg_boxsize=None, limfrac=0.5):
    """-> (mini_median, mini_std) float32 device tensors (ny/box, nx/box), NaN boxes filled
    and 3x3-median filtered (zogy.get_back)"""
    box = bkg_boxsize or settings.bkg_boxsize
    ny, nx = data.shape
    nby, nbx = ny // box, nx // box
    med = torch.empty((nby, nbx), dtype=torch.float32, device=ctx.device)
    std = torch.empty((nby, nbx), dtype=torch.float32, device=ctx.device)
    check(lib.bbx_bkg_boxstats(ctx.h, ny, nx, box, _p(data), _p(data_mask), _p(objmask), float(limfrac), _p(med), _p(std),
                               ctx.stream()), 'bbx_bkg_boxstats', ctx.h)
    for m in (med, std):
        check(lib.bbx_mini_fill_filter(ctx.h, nby, nbx, _p(m), ctx.stream()), 'bbx_mini_fill_filter', ctx.h)
    return med, std


def _bspline_weights(t):
    return np.stack([(1 - t) ** 3 / 6, (3 * t ** 3 - 6 * t ** 2 + 4) / 6, (-3 * t ** 3 + 3 * t ** 2 + 3 * t + 1) / 6,
                     t ** 3 / 6], -1)


def _axis_map(nin, nout, offset):
    o = np.arange(nout)
    cc = o * ((nin - 1) / (nout - 1)) + NPAD if nout > 1 else np.zeros(1) + NPAD
    fl = np.floor(cc).astype(np.int64)
    return (fl + offset).astype(np.int32), _bspline_weights(cc - fl)


import functools


@functools.lru_cache(maxsize=16)
def _tap_tables(nby, nbx, box, cy, cx):
    """per-row / per-column tap tables of the zoom (shape-only: cached)"""
    py, px = cy + 2 * NPAD, cx + 2 * NPAD
    fy = np.empty(nby * box, np.int32); wy = np.empty((nby * box, 4))
    fx = np.empty(nbx * box, np.int32); wx = np.empty((nbx * box, 4))
    for iy in range(nby // cy):
        f, w = _axis_map(cy, cy * box, iy * py)
        fy[iy * cy * box:(iy + 1) * cy * box], wy[iy * cy * box:(iy + 1) * cy * box] = f, w
    for ix in range(nbx // cx):
        f, w = _axis_map(cx, cx * box, ix * px)
        fx[ix * cx * box:(ix + 1) * cx * box], wx[ix * cx * box:(ix + 1) * cx * box] = f, w
    for a in (fy, wy, fx, wx):
        a.setflags(write=False)
    return fy, wy, fx, wx


def zoom_coefficients(mini, channels=None):
    """B-spline coefficients of the (edge-padded) mini image; channels=(cy, cx) boxes -> every
    channel block gets its own padded coefficient patch (interp_Xchan=False)"""
    mini = np.asarray(mini, np.float64)
    nby, nbx = mini.shape
    cy, cx = (nby, nbx) if channels is None else channels
    py, px = cy + 2 * NPAD, cx + 2 * NPAD
    coef = np.empty(((nby // cy) * py, (nbx // cx) * px))
    for iy in range(nby // cy):
        for ix in range(nbx // cx):
            blk = np.pad(mini[iy * cy:(iy + 1) * cy, ix * cx:(ix + 1) * cx], NPAD, mode='edge')
            coef[iy * py:(iy + 1) * py, ix * px:(ix + 1) * px] = ndimage.spline_filter(blk, order=3, mode='nearest',
                                                                                      output=np.float64)
    return coef


def zoom_plan(mini, box, channels=None):
    """host part of mini2back: B-spline coefficients of the (edge-padded) mini image and the
    per-row / per-column tap tables for scipy.ndimage.zoom(mini, box, order=3, mode='nearest');
    channels=(cy, cx) boxes -> every channel block gets its own padded coefficient patch
    (interp_Xchan=False)."""
    nby, nbx = np.shape(mini)
    cy, cx = (nby, nbx) if channels is None else channels
    return (zoom_coefficients(mini, channels),) + _tap_tables(nby, nbx, box, cy, cx)


def _device_taps(ctx, nby, nbx, box, cy, cx):
    """the tap tables on the context's device (kept with the context)"""
    cache = ctx.__dict__.setdefault('_zoom_taps', {})
    key = (nby, nbx, box, cy, cx)
    if key not in cache:
        cache[key] = tuple(torch.from_numpy(np.array(a)).to(ctx.device) for a in _tap_tables(*key))      # (writable copies)
    return cache[key]


SPLINE_POLE = -0.2679491924311227          # sqrt(3) - 2 correctly rounded: the constant gcc folds into scipy's ni_splines.c


def device_zoom_coefficients(ctx, mini, channels=None):
    """zoom_coefficients on the device (bbx_spline_prefilter: scipy's prefilter operation by operation, same bits):
    mini = float32 device tensor [nby, nbx] -> float64 device tensor of the padded coefficient patches"""
    import math
    nby, nbx = mini.shape
    cy, cx = (nby, nbx) if channels is None else channels
    py, px = cy + 2 * NPAD, cx + 2 * NPAD
    coef = torch.empty(((nby // cy) * py, (nbx // cx) * px), dtype=torch.float64, device=mini.device)
    check(lib.bbx_spline_prefilter(ctx.h, nby, nbx, cy, cx, NPAD, math.pow(SPLINE_POLE, py), math.pow(SPLINE_POLE, px), _p(mini), _p(coef),
                                   ctx.stream()), 'bbx_spline_prefilter', ctx.h)
    return coef


def mini2back(ctx, mini, shape, bkg_boxsize=None, interp_Xchan=True, subtract_from=None, want_bkg=True, subtract_into=None):
    """zogy.mini2back(data_mini, data_shape, order_interp=3, bkg_boxsize, interp_Xchan): full-
    frame background from the mini image; with subtract_from the same pass does `data -= bkg`
    (in place, or into the tensor subtract_into with subtract_from left as it is).  A float32 mini image (device tensor
    or numpy) never leaves the device: the B-spline prefilter runs there too (no host round trip in the frame's path);
    a float64 numpy mini image takes scipy's prefilter on the host."""
    box = bkg_boxsize or settings.bkg_boxsize
    dev = ctx.device
    nby, nbx = mini.shape
    channels = None
    if not interp_Xchan:
        if nby % settings.ny or nbx % settings.nx:
            raise ValueError('interp_Xchan=False needs a whole number of boxes per channel: mini image {}x{} over {}x{} channels'
                             .format(nby, nbx, settings.ny, settings.nx))
        channels = (nby // settings.ny, nbx // settings.nx)
    cy, cx = (nby, nbx) if channels is None else channels
    if not torch.is_tensor(mini) and np.asarray(mini).dtype != np.float32:
        d_coef = torch.from_numpy(zoom_coefficients(np.asarray(mini), channels)).to(dev)
    else:
        d_mini = mini if torch.is_tensor(mini) else push(ctx, np.ascontiguousarray(mini))
        if d_mini.dtype != torch.float32 or not d_mini.is_contiguous():
            d_mini = d_mini.to(torch.float32).contiguous()
        d_coef = device_zoom_coefficients(ctx, d_mini, channels)
    cshape = tuple(d_coef.shape)
    d_fy, d_wy, d_fx, d_wx = _device_taps(ctx, nby, nbx, box, cy, cx)
    ny, nx = shape
    if subtract_into is not None:
        check(lib.bbx_spline_zoom_sub(ctx.h, ny, nx, _p(d_coef), cshape[0], cshape[1], _p(d_fy), _p(d_wy), _p(d_fx),
                                      _p(d_wx), _p(subtract_from), _p(subtract_into), ctx.stream()), 'bbx_spline_zoom_sub', ctx.h)
        return None
    bkg = torch.empty((ny, nx), dtype=torch.float32, device=dev) if want_bkg else None
    check(lib.bbx_spline_zoom(ctx.h, ny, nx, _p(d_coef), cshape[0], cshape[1], _p(d_fy), _p(d_wy), _p(d_fx),
                              _p(d_wx), _p(subtract_from), _p(bkg), ctx.stream()), 'bbx_spline_zoom', ctx.h)
    return bkg


class MiniImage:
    """A mini (per-box) image in the form the kernels read it at frame pixels (bbx_zogy_frame_mini, bbx_psf_optflux_mini):
    the B-spline coefficients of its edge-padded patches on the device + the geometry of zogy.mini2back(mini, shape,
    bkg_boxsize, interp_Xchan).  The full-frame image is never made; frame() makes it for callers that need one."""

    def __init__(self, ctx, mini, box, interp_Xchan=True):
        nby, nbx = mini.shape
        self.channels = None
        if not interp_Xchan:
            if nby % settings.ny or nbx % settings.nx:
                raise ValueError('interp_Xchan=False needs a whole number of boxes per channel: mini image {}x{} over {}x{} channels'
                                 .format(nby, nbx, settings.ny, settings.nx))
            self.channels = (nby // settings.ny, nbx // settings.nx)
        cy, cx = (nby, nbx) if self.channels is None else self.channels
        if torch.is_tensor(mini) or np.asarray(mini).dtype == np.float32:
            d_mini = mini if torch.is_tensor(mini) else push(ctx, np.ascontiguousarray(mini))
            if d_mini.dtype != torch.float32 or not d_mini.is_contiguous():
                d_mini = d_mini.to(torch.float32).contiguous()
            self.coef = device_zoom_coefficients(ctx, d_mini, self.channels)
        else:
            self.coef = torch.from_numpy(zoom_coefficients(np.asarray(mini), self.channels)).to(ctx.device)
        self.box, self.shape = int(box), (nby * int(box), nbx * int(box))
        self.c = _SplineImage(self.coef.data_ptr(), nby, nbx, cy, cx, int(box), NPAD)

    def ref(self):
        return C.byref(self.c)

    def frame(self, ctx):
        """the full-frame image (bbx_spline_zoom of the same coefficients)"""
        nby, nbx = self.c.nby, self.c.nbx
        d_fy, d_wy, d_fx, d_wx = _device_taps(ctx, nby, nbx, self.box, self.c.cy, self.c.cx)
        out = torch.empty(self.shape, dtype=torch.float32, device=ctx.device)
        check(lib.bbx_spline_zoom(ctx.h, self.shape[0], self.shape[1], _p(self.coef), self.coef.shape[0], self.coef.shape[1], _p(d_fy), _p(d_wy),
                                  _p(d_fx), _p(d_wx), None, _p(out), ctx.stream()), 'bbx_spline_zoom', ctx.h)
        return out


def mini_path_supported(shape, size, border, box, *minis):
    """can bbx_zogy_frame_mini read these mini images for a frame of this geometry (aligned groups of four pixels)?"""
    ny, nx = shape
    if size % 4 or border % 4 or nx % 4 or (size + 2 * border) % 4:
        return False
    for m in minis:
        if m.shape != (ny, nx) or (m.c.cx * m.box) % 4:
            return False
    return True


# ---- ZOGY ---------------------------------------------------------------------------------
def cut_subimages(ctx, img, size=None, border=None):
    size = size or settings.subimage_size
    border = settings.subimage_border if border is None else border
    ny, nx = img.shape
    L = size + 2 * border
    nsub = (ny // size) * (nx // size)
    subs = torch.empty((nsub, L, L), dtype=torch.float32, device=ctx.device)
    check(lib.bbx_cut_subimages(ctx.h, ny, nx, size, border, _p(img), _p(subs), ctx.stream()), 'bbx_cut_subimages', ctx.h)
    return subs


def stitch_subimages(ctx, subs, shape, size=None, border=None):
    size = size or settings.subimage_size
    border = settings.subimage_border if border is None else border
    ny, nx = shape
    img = torch.empty((ny, nx), dtype=torch.float32, device=ctx.device)
    check(lib.bbx_stitch_subimages(ctx.h, ny, nx, size, border, _p(subs), _p(img), ctx.stream()), 'bbx_stitch_subimages', ctx.h)
    return img


def run_zogy(ctx, N, R, Pn, Pr, Vn, Vr, scal):
    """batched run_ZOGY: all inputs [nsub, L, L] float32 device tensors; scal [nsub, 6] =
    (sigma_n, sigma_r, f_n, f_r, dx, dy) -> D, S, Scorr, Fpsf, Fpsferr"""
    nsub, L, _ = N.shape
    scal = np.ascontiguousarray(scal, dtype=np.float32)
    assert scal.shape == (nsub, 6)
    outs = [torch.empty_like(N) for _ in range(5)]
    check(lib.bbx_zogy_subimages(ctx.h, L, nsub, _p(N), _p(R), _p(Pn), _p(Pr), _p(Vn), _p(Vr),
                                 scal.ctypes.data_as(C.POINTER(C.c_float)), *[_p(o) for o in outs], ctx.stream()),
          'bbx_zogy_subimages', ctx.h)
    return outs


def frame_path_supported(L):
    """sub-image sides the hand-written FFT path (bbx_zogy_frame) is built for; BBX_ZOGY_ROCFFT=1
    forces the rocFFT path (bbx_zogy_subimages) everywhere"""
    import os
    return bool(lib.bbx_zogy_frame_supported(int(L))) and not os.environ.get('BBX_ZOGY_ROCFFT')


def zogy_frame_outputs(new, want_S=False):
    """the frames bbx_zogy_frame fills: D, S (or None), Scorr, Fpsf, Fpsferr"""
    return [torch.empty_like(new) if (k != 1 or want_S) else None for k in range(5)]


def run_zogy_frame(ctx, new, ref, sig_new, sig_ref, psf_n, psf_r, scal, size, border, want_S=False, outs=None):
    """ZOGY of whole frames (bbx_zogy_frame): background-subtracted frames + sigma images + PSF
    stamps [nsub, S, S] -> D, S (or None), Scorr, Fpsf, Fpsferr full frames.  sig_new, sig_ref: frames, or both
    MiniImage (bbx_zogy_frame_mini: the sigma maps are read off their mini images, no frames exist)"""
    ny, nx = new.shape
    nsub = (ny // size) * (nx // size)
    scal = np.ascontiguousarray(scal, dtype=np.float32)
    assert scal.shape == (nsub, 6) and psf_n.shape[0] == nsub and psf_r.shape[0] == nsub
    S = int(psf_n.shape[1])
    outs = outs or zogy_frame_outputs(new, want_S)
    if isinstance(sig_new, MiniImage):
        check(lib.bbx_zogy_frame_mini(ctx.h, ny, nx, int(size), int(border), _p(new), _p(ref), sig_new.ref(), sig_ref.ref(),
                                      _p(psf_n.contiguous()), _p(psf_r.contiguous()), S, scal.ctypes.data_as(C.POINTER(C.c_float)),
                                      *[_p(o) for o in outs], ctx.stream()), 'bbx_zogy_frame_mini', ctx.h)
        return outs
    check(lib.bbx_zogy_frame(ctx.h, ny, nx, int(size), int(border), _p(new), _p(ref), _p(sig_new), _p(sig_ref),
                             _p(psf_n.contiguous()), _p(psf_r.contiguous()), S, scal.ctypes.data_as(C.POINTER(C.c_float)),
                             *[_p(o) for o in outs], ctx.stream()), 'bbx_zogy_frame', ctx.h)
    return outs


def psf_optflux(ctx, D, V, psfs, ys, xs, v_is_sigma=False):
    """zogy.get_psfoptflux at integer positions -> (flux, fluxerr) float32 device tensors; with v_is_sigma
    V is the sigma image of the background-subtracted frame D -- a frame, or a MiniImage read at the stamp pixels --
    and the variance max(D, 0) + sigma^2 is formed at the stamp pixels only"""
    nsrc, S, _ = psfs.shape
    dev = ctx.device
    d_ys, d_xs = push(ctx, np.asarray(ys, np.int32), np.asarray(xs, np.int32))
    flux = torch.empty(nsrc, dtype=torch.float32, device=dev)
    err = torch.empty(nsrc, dtype=torch.float32, device=dev)
    ny, nx = D.shape
    if isinstance(V, MiniImage):
        check(lib.bbx_psf_optflux_mini(ctx.h, ny, nx, _p(D), V.ref(), _p(psfs), S, nsrc, _p(d_ys), _p(d_xs), _p(flux), _p(err), ctx.stream()),
              'bbx_psf_optflux_mini', ctx.h)
        return flux, err
    fn = lib.bbx_psf_optflux_sigma if v_is_sigma else lib.bbx_psf_optflux
    check(fn(ctx.h, ny, nx, _p(D), _p(V), _p(psfs), S, nsrc, _p(d_ys), _p(d_xs), _p(flux), _p(err), ctx.stream()), 'bbx_psf_optflux', ctx.h)
    return flux, err


def psf_poly_terms(x, y, polzero, polscal, poldeg):
    """PSFEx polynomial terms of the source positions (pixel coordinates x, y; one
    polynomial group over (x, y)): x' = (x - polzero[0]) / polscal[0], same for y; order
    1, x', x'^2, .., y', x'y', .., y'^2, .. (y power outer, x power inner), float32.
    -> array [nsrc, (poldeg+1)(poldeg+2)/2]"""
    xn = ((np.asarray(x, np.float64) - polzero[0]) / polscal[0]).astype(np.float32)
    yn = ((np.asarray(y, np.float64) - polzero[1]) / polscal[1]).astype(np.float32)
    cols = []
    for j in range(poldeg + 1):
        for i in range(poldeg + 1 - j):
            cols.append((xn ** np.float32(i)) * (yn ** np.float32(j)) if (i or j) else np.ones_like(xn))
    return np.stack(cols, axis=1).astype(np.float32)


def resample_psf_basis(basis, psf_samp):
    """PSFEx tabulates its model every PSF_SAMP image pixels (automatic sampling: usually != 1).  zogy.get_psf_ima
    (buildref.py:3357-3366 passes psf_samp) resamples the model image to image pixels: psf_size =
    ceil(S_config * psf_samp) made odd, scipy.ndimage.zoom by psf_size / S_config [EXT: order 2, mode 'nearest';
    oracle/zogy_core.get_psf_ima].  The resampling is linear, so it is applied once to every basis plane (host,
    ncoef small images) and the per-source contraction (bbx_psf_model) then works on image pixels.
    -> float32 [ncoef, psf_size, psf_size]"""
    basis = np.asarray(basis, np.float64)
    s_cfg = basis.shape[1]
    size = int(np.ceil(s_cfg * float(psf_samp)))
    size += 1 - size % 2
    if size == s_cfg:
        return basis.astype(np.float32)
    out = np.stack([ndimage.zoom(b, size / s_cfg, order=2, mode='nearest') for b in basis])
    if out.shape[1:] != (size, size):
        raise ValueError('PSF resampling gave {} instead of {}'.format(out.shape[1:], (size, size)))
    return out.astype(np.float32)


def psf_model_stamps(ctx, basis, x, y, polzero, polscal, poldeg, normalize=True):
    """PSF stamp of every source from a PSFEx model: basis [ncoef, S, S] float32 device
    tensor (PSF_MASK), positions x, y (host arrays).  The contraction runs on the f32 MFMA
    (bbx_psf_model).  -> device tensor [nsrc, S, S]; normalize: each stamp sums to one"""
    terms = torch.from_numpy(psf_poly_terms(x, y, polzero, polscal, poldeg)).to(ctx.device)
    ncoef, S = basis.shape[0], basis.shape[1]
    if terms.shape[1] != ncoef:
        raise ValueError('basis has %d planes, polynomial degree %d needs %d' % (ncoef, poldeg, terms.shape[1]))
    nsrc = terms.shape[0]
    out = torch.empty((nsrc, S, S), dtype=torch.float32, device=ctx.device)
    check(lib.bbx_psf_model(ctx.h, nsrc, ncoef, S * S, _p(terms), _p(basis.contiguous()), _p(out), ctx.stream()),
          'bbx_psf_model', ctx.h)
    if normalize:
        out /= out.sum(dim=(1, 2), keepdim=True)
    return out


def find_transients(ctx, Scorr, nsigma=None, max_out=100000):
    """connected regions of |Scorr| >= T-NSIGMA -> sorted list of (y, x, Scorr peak)"""
    ys, xs, val = find_peaks_arrays(ctx, Scorr, nsigma, max_out)
    return list(zip(ys.tolist(), xs.tolist(), val.tolist()))


def find_peaks_enqueue(ctx, Scorr, nsigma=None, max_out=100000):
    """queue bbx_find_peaks -> the device tensors find_peaks_collect reads (host work in between overlaps the search)"""
    nsigma = settings.transient_nsigma if nsigma is None else nsigma
    ny, nx = Scorr.shape
    dev = ctx.device
    yx = torch.empty((max_out, 2), dtype=torch.int32, device=dev)
    val = torch.empty(max_out, dtype=torch.float32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib.bbx_find_peaks(ctx.h, ny, nx, _p(Scorr), float(nsigma), max_out, _p(yx), _p(val), _p(cnt), ctx.stream()),
          'bbx_find_peaks', ctx.h)
    return yx, val, cnt, max_out


def find_peaks_collect(ctx, pending):
    """-> arrays (y int32, x int32, peak float32), sorted by (y, x)"""
    yx, val, cnt, max_out = pending
    if max_out * 12 <= (2 << 20):
        # the count and the whole list (1.2 MB at the default capacity) in ONE copy back and one host wait: asking for the
        # count first costs a second round trip -- with the stream's other work in front of it each time
        c, yx, val = fetch(ctx, cnt, yx, val, check_device_errors=True)
        n = min(int(c[0]), max_out)
        yx, val = yx[:n], val[:n]
    else:
        n = min(int(fetch(ctx, cnt, check_device_errors=True)[0]), max_out)
        yx, val = fetch(ctx, yx[:n], val[:n])
    order = np.lexsort((yx[:, 1], yx[:, 0])) if n else np.zeros(0, int)
    return yx[order, 0], yx[order, 1], val[order]


def find_peaks_arrays(ctx, Scorr, nsigma=None, max_out=100000):
    """the same as arrays (y int32, x int32, peak float32), sorted by (y, x): no per-source Python work"""
    return find_peaks_collect(ctx, find_peaks_enqueue(ctx, Scorr, nsigma, max_out))


def embed_psfs(ctx, stamps, L):
    """PSF stamps [nsub, S, S] (unit sum, centre at S//2) -> [nsub, L, L] centred on pixel [0,0]"""
    nsub, S, _ = stamps.shape
    out = torch.empty((nsub, L, L), dtype=torch.float32, device=ctx.device)
    check(lib.bbx_embed_psf(ctx.h, nsub, S, L, _p(stamps), _p(out), ctx.stream()), 'bbx_embed_psf', ctx.h)
    return out


def variance(ctx, data_bkgsub, bkg_std):
    v = torch.empty_like(data_bkgsub)
    check(lib.bbx_variance(ctx.h, data_bkgsub.numel(), _p(data_bkgsub), _p(bkg_std), _p(v), ctx.stream()), 'bbx_variance', ctx.h)
    return v


def remap_mini(mini, grid, shape_in, box, step):
    """a mini (per-box) image of the reference frame sampled on the new frame's box centres:
    bilinear interpolation of [mini] at the reference-frame positions that the projection
    lattice [grid] (coadd.projection_grid: [gny][gnx][2] = x, y input position of the output
    pixels (j*step, i*step)) assigns to the centres of the new frame's boxes.  Host, float64:
    176 x 176 values."""
    mini = np.asarray(mini, np.float64)
    nby, nbx = mini.shape
    cy = (np.arange(nby) + 0.5) * box - 0.5
    cx = (np.arange(nbx) + 0.5) * box - 0.5
    gy, gx = cy / step, cx / step
    j0 = np.clip(np.floor(gy).astype(int), 0, grid.shape[0] - 2); fy = (gy - j0)[:, None]
    i0 = np.clip(np.floor(gx).astype(int), 0, grid.shape[1] - 2); fx = (gx - i0)[None, :]
    g00, g01 = grid[j0][:, i0], grid[j0][:, i0 + 1]
    g10, g11 = grid[j0 + 1][:, i0], grid[j0 + 1][:, i0 + 1]
    pos = ((1 - fy) * (1 - fx))[..., None] * g00 + ((1 - fy) * fx)[..., None] * g01 + \
          (fy * (1 - fx))[..., None] * g10 + (fy * fx)[..., None] * g11
    # position in box units of the reference frame's mini image (box centres at k + 0.5)
    by = np.clip((pos[..., 1] + 0.5) / box - 0.5, 0, shape_in[0] // box - 1)
    bx = np.clip((pos[..., 0] + 0.5) / box - 0.5, 0, shape_in[1] // box - 1)
    y0 = np.clip(np.floor(by).astype(int), 0, mini.shape[0] - 2); wy = by - y0
    x0 = np.clip(np.floor(bx).astype(int), 0, mini.shape[1] - 2); wx = bx - x0
    out = (1 - wy) * (1 - wx) * mini[y0, x0] + (1 - wy) * wx * mini[y0, x0 + 1] + \
          wy * (1 - wx) * mini[y0 + 1, x0] + wy * wx * mini[y0 + 1, x0 + 1]
    return out.astype(np.float32)


def subimage_psfs(ctx, psf, nsy, nsx, size):
    """PSF stamps of the sub-images [nsub, S, S] (unit sum) from either such a tensor, a single
    stamp [S, S], or a PSFEx model dict(basis=[ncoef,S,S] device tensor, polzero, polscal, poldeg)
    evaluated at the sub-image centres on the f32 MFMA (zogy.get_psf_ima at the tile centre)"""
    nsub = nsy * nsx
    if isinstance(psf, dict):
        yc = (np.arange(nsy) + 0.5) * size + 0.5          # FITS pixel coordinates of the tile centres
        xc = (np.arange(nsx) + 0.5) * size + 0.5
        yy, xx = np.meshgrid(yc, xc, indexing='ij')
        return psf_model_stamps(ctx, psf['basis'], xx.ravel(), yy.ravel(), psf['polzero'], psf['polscal'], psf['poldeg'])
    if psf.dim() == 2:
        return psf.unsqueeze(0).expand(nsub, -1, -1).contiguous()
    if psf.shape[0] != nsub:
        raise ValueError('need one PSF stamp per sub-image ({}), got {}'.format(nsub, psf.shape[0]))
    return psf.contiguous()


def source_psfs(ctx, psf, sub_psfs, ys, xs, nsx, size):
    """unit-sum PSF stamp of every source: the PSFEx model at the source position, or the stamp
    of the sub-image the source falls in"""
    if isinstance(psf, dict):
        return psf_model_stamps(ctx, psf['basis'], np.asarray(xs) + 1.0, np.asarray(ys) + 1.0, psf['polzero'],
                                psf['polscal'], psf['poldeg'])
    k = push(ctx, ((np.asarray(ys) // size) * nsx + (np.asarray(xs) // size)).astype(np.int64))
    return sub_psfs.index_select(0, k).contiguous()


def frame_clipped_stats_enqueue(ctx, img, mask=None, step=8):
    """queue bbx_frame_clipped_stats of a contiguous float32 frame -> device tensor [8] (n, median, mean, sigma, ...)"""
    if img.dim() != 2 or img.dtype != torch.float32 or not img.is_contiguous():
        raise ValueError('contiguous 2-D float32 frame expected')
    if mask is not None and (mask.dtype != torch.uint8 or mask.shape != img.shape or not mask.is_contiguous()):
        raise ValueError('mask: contiguous uint8 of the frame shape expected')
    ny, nx = img.shape
    out = torch.empty(8, dtype=torch.float64, device=img.device)
    check(lib.bbx_frame_clipped_stats(ctx.h, ny, nx, _p(img), _p(mask), int(step), 3.0, 5, 1,
                                      _p(out), ctx.stream()), 'bbx_frame_clipped_stats', ctx.h)
    return out


def frame_clipped_stats(ctx, img, mask=None, step=8):
    """sigma_clipped_stats (3 sigma, 5 iterations, centre = exact median, mask_value 0) of a frame -> (median,
    std) as zogy reports them in Z-SCMED / Z-SCSTD / Z-FPEMED / Z-FPESTD.  zogy takes these header
    statistics from a random subset of the pixels; here the subset is the regular lattice of
    every [step]-th pixel in both axes (deterministic; 1.7 10^6 samples of a 10560^2 frame)."""
    st = fetch(ctx, frame_clipped_stats_enqueue(ctx, img, mask, step))
    return float(st[1]), float(st[3])


class StreamGate:
    """Lets one stream at a time run a section on the GPU: a section starts when the previous one (of any
    stream) has finished.  bbx_zogy_frame's kernels each fill the whole GPU; two lanes running them side
    by side only slice each other's time.  With priority=True the sections run on ONE high-priority stream of the
    gate's own (in order: that is the gate), between a wait for the caller's stream and a wait of the caller's stream:
    the short kernels of the other lanes then take the CUs a section leaves free instead of an equal share of all of
    them.  Tensors a section is to fill must be allocated before it (they belong to the caller's stream)."""

    def __init__(self, device=None, priority=False):
        import threading
        self.lock = threading.Lock()
        self.last = None
        self.hp = torch.cuda.Stream(device=device, priority=-1) if priority else None
        self._cm = self._cur = None

    def __enter__(self):
        self.lock.acquire()
        cur = torch.cuda.current_stream()
        if self.hp is not None:
            self.hp.wait_stream(cur)
            self._cur, self._cm = cur, torch.cuda.stream(self.hp)
            self._cm.__enter__()
        elif self.last is not None:
            cur.wait_event(self.last)
        return self

    def __exit__(self, *exc):
        if self.hp is not None:
            self._cm.__exit__(*exc)
            self._cur.wait_stream(self.hp)
            self._cm = self._cur = None
        else:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.last = ev
        self.lock.release()
        return False


class _NoGate:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def optimal_subtraction(ctx, *args, **kw):
    try:
        return _optimal_subtraction(ctx, *args, **kw)
    finally:
        # the candidate-list handoffs (zoom -> catalogue search, final ZOGY kernel -> transient search) are keyed on the
        # frames' addresses inside the context: whatever way this call ends, none of that may outlive it (the caching
        # allocator hands the same address to the next frame)
        lib.bbx_zoom_candidates(ctx.h, None, 0.0)
        lib.bbx_zogy_candidates(ctx.h, 0.0)


def _optimal_subtraction(ctx, new, ref, new_mask, ref_mask, psf_new, psf_ref, fratio=1.0, dx=0.0, dy=0.0,
                         subimage_size=None, subimage_border=None, bkg_boxsize=None, nsigma=None,
                         ref_is_bkgsub=False, ref_bkg_std_mini=None, ref_grid=None, ref_grid_step=32,
                         cat_extract=False, cat_nsigma=5.0, trans_extract=True, frame_stats=True, max_sources=200000,
                         zogy_gate=None, ref_bkg_std=None, sigma_frames=False):
    """The numerical core of zogy.optimal_subtraction(new_fits, ref_fits, ...) (call sites
    blackbox.py:2350-2354 new-only, 2460-2465 new + ref) on device tensors: background mesh +
    subtraction, variance images, [remapping of the reference to the new frame's grid],
    sub-image ZOGY, stitching, transient candidates with PSF fluxes, [full-source catalogue by
    PSF-weighted optimal photometry].
      new, new_mask : reduced frame (float32, e-) and its mask
      fratio, dx, dy: flux ratio new / ref and the astrometric scatter, one number or one per sub-image
      ref, ref_mask : reference frame or None (new-only mode, trans_extract False or no ref:
                      the 2350-2354 branch): then only the background products and the catalogue
      ref_is_bkgsub : the reference is a background-subtracted co-add (buildref product);
                      ref_bkg_std_mini: its `_bkg_std_mini` image (else measured here);
                      ref_bkg_std: what an earlier call made from it (res['bkg_std_ref']: a MiniImage, or the
                      full-frame sigma image), when the caller keeps it for many frames of the same field
      sigma_frames  : True: the two sigma images are made as full frames (rounds 1-4; bbx_zogy_frame) instead of
                      being read off their mini images inside the kernels (bbx_zogy_frame_mini)
      ref_grid      : projection lattice (coadd.projection_grid) when the reference lives on
                      another pixel grid: it is remapped with the LANCZOS3 kernel (zogy runs SWarp)
      psf_new/ref   : PSF stamps [nsub, S, S] / [S, S] (unit sum) or a PSFEx model dict
                      (subimage_psfs) -- running PSFEx itself is out of scope
    -> dict(D, Scorr, Fpsf, Fpsferr, bkg_mini_new, bkg_std_mini_new, ..., transients, catalog,
            header (= header_new additions), header_trans)"""
    size = subimage_size or settings.subimage_size
    border = settings.subimage_border if subimage_border is None else subimage_border
    box = bkg_boxsize or settings.bkg_boxsize
    ny, nx = new.shape
    L = size + 2 * border
    res, hdr, hdr_t = {}, {}, {}
    nsy, nsx = ny // size, nx // size
    nsub = nsy * nsx
    have_ref = ref is not None and trans_extract

    # ---- new frame: mesh, subtraction, sigma image, variance
    mini, mini_std = get_back(ctx, new, new_mask, bkg_boxsize=box)
    work = torch.empty_like(new)
    want_cat = cat_extract and psf_new is not None
    if want_cat:
        # the kernel that writes the background-subtracted frame lists the pixels above cat_nsigma x S-BKGSTD for the
        # catalogue's peak search; S-BKGSTD = median of the sigma mini image, taken on the device for that
        d_sstd = torch.empty(1, dtype=torch.float32, device=ctx.device)
        check(lib.bbx_mini_median(ctx.h, mini_std.numel(), _p(mini_std), _p(d_sstd), ctx.stream()), 'bbx_mini_median', ctx.h)
        check(lib.bbx_zoom_candidates(ctx.h, _p(d_sstd), float(cat_nsigma)), 'bbx_zoom_candidates', ctx.h)
    mini2back(ctx, mini, (ny, nx), bkg_boxsize=box, interp_Xchan=True, subtract_from=new, subtract_into=work)
    # the sigma image of the new frame: read off its mini image by the kernels that need it (catalogue photometry, the cut
    # into sub-images) where the geometry allows, a frame otherwise
    import os
    use_mini = frame_path_supported(L) and not sigma_frames and not os.environ.get('BBX_SIGMA_FRAMES')     # (the switch: A/B timing)
    bstd = None
    if use_mini:
        try:
            bstd = MiniImage(ctx, mini_std, box, interp_Xchan=False)
            use_mini = mini_path_supported((ny, nx), size, border, box, bstd)
        except ValueError:
            use_mini = False
    if not use_mini:
        bstd = mini2back(ctx, mini_std, (ny, nx), bkg_boxsize=box, interp_Xchan=False)
    Vn = None                                                        # variance image: only where a consumer needs it
    res['bkg_mini_new'], sdn = fetch(ctx, mini, mini_std)
    res['bkg_std_mini_new'] = sdn
    hdr['BKG-SIZE'] = (box, '[pix] background boxsize used')
    hdr['BKG-SUB'] = (False, 'sky background was subtracted?')          # the _red product keeps its sky
    # zogy's per-channel background correction factors (BKG-CF1..16, BKG-FDEG, BKG-FC0: blackbox.py:3061-3066, all None_OK)
    # are [EXT] without a source in the reference tree: not applied here, and the header says so
    hdr['BKG-CORR'] = (False, 'channel background correction applied?')
    hdr['S-BKGSTD'] = (float(np.median(sdn)), '[e-] sigma (STD) background full-frame image')
    res['data_bkgsub'], res['bkg_std'] = work, bstd
    bs = size // box if size % box == 0 else None

    def tile_medians(a):
        """median of the mini image over the boxes of each sub-image (or over all of it)"""
        if bs and a.shape == (nsy * bs, nsx * bs):
            return np.median(a.reshape(nsy, bs, nsx, bs).transpose(0, 2, 1, 3).reshape(nsub, bs * bs), axis=1)
        if bs:
            return np.asarray([np.median(a[(k // nsx) * bs:(k // nsx + 1) * bs, (k % nsx) * bs:(k % nsx + 1) * bs]) for k in range(nsub)])
        return np.full(nsub, np.median(a))
    scal_n = None

    def host_side_meanwhile():
        """host work that needs nothing from the device: done while a search runs there"""
        hdr['S-BKG'] = (float(np.median(res['bkg_mini_new'])), '[e-] median background full-frame image')
        return tile_medians(sdn) if have_ref else None

    sub_pn = subimage_psfs(ctx, psf_new, nsy, nsx, size) if psf_new is not None else None

    # ---- full-source catalogue (a17): peaks of the background-subtracted frame above
    # cat_nsigma x S-BKGSTD, PSF-weighted optimal flux at each (zogy.get_psfoptflux)
    res['catalog'] = None
    if cat_extract and sub_pn is not None:
        thr = float(cat_nsigma) * hdr['S-BKGSTD'][0]
        if np.isfinite(thr) and thr > 0:
            pending = find_peaks_enqueue(ctx, work, thr, max_out=max_sources)
            scal_n = host_side_meanwhile()
            ys, xs, pk = find_peaks_collect(ctx, pending)
        else:                                                        # no usable noise level: nothing is significant
            lib.bbx_zoom_candidates(ctx.h, None, 0.0)
            scal_n = host_side_meanwhile()
            ys = xs = np.zeros(0, np.int32); pk = np.zeros(0, np.float32)
        keep = pk > 0
        ys, xs, pk = ys[keep], xs[keep], pk[keep]
        if ys.size:
            # peaks on masked pixels are dropped; their mask values come back with the fluxes (one host wait: the photometry
            # of the few masked ones is made and thrown away)
            d_ys, d_xs = push(ctx, ys.astype(np.int64), xs.astype(np.int64))
            d_mk = new_mask[d_ys, d_xs]
            stamps = source_psfs(ctx, psf_new, sub_pn, ys, xs, nsx, size)
            f, e = psf_optflux(ctx, work, bstd, stamps, ys, xs, v_is_sigma=True)        # (bstd: frame or MiniImage)
            mk, f, e = fetch(ctx, d_mk, f, e)
            ok = mk == 0
            ys, xs, pk, f, e = ys[ok], xs[ok], pk[ok], f[ok], e[ok]
        else:
            f = e = np.zeros(0, np.float32)
        peaks = ys
        res['catalog'] = dict(Y_POS=ys.astype(np.float32) + 1, X_POS=xs.astype(np.float32) + 1,
                              E_FLUX_PEAK=pk.astype(np.float32), E_FLUX_OPT=f, E_FLUXERR_OPT=e,
                              SNR_OPT=np.where(e > 0, f / np.where(e > 0, e, 1), 0).astype(np.float32))
        hdr['NOBJECTS'] = (len(peaks), 'number of objects detected')
    if 'S-BKG' not in hdr:
        scal_n = host_side_meanwhile()
    if not have_ref:
        hdr['Z-P'] = (False, 'successfully processed by ZOGY?')
        res['header'], res['header_new'], res['header_trans'], res['transients'] = _HeaderView(hdr, hdr_t), hdr, hdr_t, []
        return res

    # ---- reference frame on the new frame's grid: background-subtracted image + sigma image
    rny, rnx = ref.shape
    if ref_is_bkgsub:
        rwork = ref
        if ref_bkg_std_mini is None:
            _, rstd_mini = get_back(ctx, ref, ref_mask, bkg_boxsize=box)
            sdr = fetch(ctx, rstd_mini)
        else:
            sdr = np.asarray(ref_bkg_std_mini, np.float32)
    else:
        rmini, rstd_mini = get_back(ctx, ref, ref_mask, bkg_boxsize=box)
        rwork = torch.empty_like(ref)
        mini2back(ctx, rmini, (rny, rnx), bkg_boxsize=box, interp_Xchan=True, subtract_from=ref, subtract_into=rwork)
        res['bkg_mini_ref'], sdr_meas = fetch(ctx, rmini, rstd_mini)
        sdr = sdr_meas if ref_bkg_std_mini is None else np.asarray(ref_bkg_std_mini, np.float32)
    res['bkg_std_mini_ref'] = sdr
    if ref_grid is not None:
        from . import coadd
        ones = torch.ones_like(rwork)
        rwork, _ = coadd.resample(ctx, rwork, ones, ref_grid, (ny, nx), 1.0, ref_grid_step)
        sdr = remap_mini(sdr, np.asarray(ref_grid), (rny, rnx), box, ref_grid_step)
    elif (rny, rnx) != (ny, nx):
        raise ValueError('reference frame of another shape needs ref_grid')
    # co-added reference: no channel structure in its noise -> interpolation across the frame
    sub_pr = subimage_psfs(ctx, psf_ref, nsy, nsx, size)
    frame_path = frame_path_supported(L) and sub_pn.shape[1] == sub_pr.shape[1]
    if ref_bkg_std is not None and ref_grid is None and tuple(ref_bkg_std.shape) == (ny, nx) and isinstance(ref_bkg_std, MiniImage) == use_mini:
        rbstd = ref_bkg_std
    elif use_mini and frame_path:
        rbstd = MiniImage(ctx, np.asarray(sdr, np.float32), box, interp_Xchan=True)
        if not mini_path_supported((ny, nx), size, border, box, rbstd):
            rbstd = rbstd.frame(ctx)
    else:
        rbstd = mini2back(ctx, sdr, (ny, nx), bkg_boxsize=box, interp_Xchan=True)
    if isinstance(bstd, MiniImage) and not (frame_path and isinstance(rbstd, MiniImage)):
        bstd = res['bkg_std'] = bstd.frame(ctx)                   # the other side needs frames: both as frames
        if isinstance(rbstd, MiniImage):
            rbstd = rbstd.frame(ctx)
    res['ref_bkgsub'], res['bkg_std_ref'] = rwork, rbstd
    hdr_t['S-BKGSTDR'] = (float(np.median(sdr)), '[e-] sigma (STD) background reference image')

    # ---- sub-images
    scal = np.zeros((nsub, 6), np.float32)
    scal[:, 0], scal[:, 1] = scal_n, tile_medians(sdr)
    # fratio, dx, dy: one number for the frame or one per sub-image (zogy measures them per sub-image from the matched stars)
    fr = np.broadcast_to(np.asarray(fratio, np.float64), (nsub,))
    scal[:, 2], scal[:, 3] = 1.0, np.where(fr != 0, 1.0 / np.where(fr != 0, fr, 1.0), 1.0)
    scal[:, 4], scal[:, 5] = np.broadcast_to(np.asarray(dx, np.float64), (nsub,)), np.broadcast_to(np.asarray(dy, np.float64), (nsub,))
    if frame_path:
        # hand-written FFT path: cut, variance images, ZOGY and stitching in one library call
        outs = zogy_frame_outputs(work)                           # allocated on the caller's stream, filled inside the gate
        sub_pn, sub_pr = sub_pn.contiguous(), sub_pr.contiguous()
        nsig_cand = float(settings.transient_nsigma if nsigma is None else nsigma)

        def zogy_frame_call():
            # the kernel that writes Scorr lists the pixels above the transient threshold for the peak search below
            check(lib.bbx_zogy_candidates(ctx.h, nsig_cand), 'bbx_zogy_candidates', ctx.h)
            with (zogy_gate or _NoGate()):
                return run_zogy_frame(ctx, work, rwork, bstd, rbstd, sub_pn, sub_pr, scal, size, border, outs=outs)
        D, _, Scorr, Fpsf, Fpsferr = zogy_frame_call()
        res['D'], res['Scorr'], res['Fpsf'], res['Fpsferr'] = D, Scorr, Fpsf, Fpsferr
    else:
        Vn = Vn if Vn is not None else variance(ctx, work, bstd)
        Vr = variance(ctx, rwork, rbstd)
        subs = [cut_subimages(ctx, a, size, border) for a in (work, rwork, Vn, Vr)]
        Pn, Pr = embed_psfs(ctx, sub_pn, L), embed_psfs(ctx, sub_pr, L)
        D, S, Scorr, Fpsf, Fpsferr = run_zogy(ctx, subs[0], subs[1], Pn, Pr, subs[2], subs[3], scal)
        del subs, Pn, Pr, S, Vr
        for name, a in (('D', D), ('Scorr', Scorr), ('Fpsf', Fpsf), ('Fpsferr', Fpsferr)):
            res[name] = stitch_subimages(ctx, a, (ny, nx), size, border)
    del D, Scorr, Fpsf, Fpsferr

    # ---- transient candidates: regions of |Scorr| >= T-NSIGMA, flux = Fpsf at the peak
    nsig = settings.transient_nsigma if nsigma is None else nsigma
    for attempt in (0, 1):
        try:
            # (the first host wait behind bbx_zogy_frame: its device-side checks surface here)
            tys, txs, tsc = find_peaks_arrays(ctx, res['Scorr'], nsig)
            ntrans = int(tys.size)
        except _lib_BBXError as e:
            if e.code == BBX_ERR_PSFWIN and frame_path and attempt == 0:
                # the matched-filter kernels of these PSFs do not fit their row window (include/bbx.h, BBX_OPT_ZOGY_KWIN_OFF):
                # V(S) of this call is off by more than the tolerance.  Once more on all rows -- the frame keeps its subtraction
                check(lib.bbx_set_option(ctx.h, BBX_OPT_ZOGY_KWIN_OFF, 1), 'bbx_set_option', ctx.h)
                try:
                    zogy_frame_call()
                finally:
                    check(lib.bbx_set_option(ctx.h, BBX_OPT_ZOGY_KWIN_OFF, 0), 'bbx_set_option', ctx.h)
                hdr_t['Z-KWIN'] = (False, 'ZOGY matched-filter kernels on a row window?')
                continue
            if e.code != BBX_ERR_OVERFLOW:
                raise
            # more significant pixels than the candidate list holds (a failed subtraction: wrong
            # reference, gross misalignment): the images stand, the candidate table stays empty
            tys = txs = np.zeros(0, np.int32); tsc = np.zeros(0, np.float32)
            ntrans = 'None'
        break
    # the fluxes at the candidates and the frame statistics of the header: queued together, one copy back, one host wait
    d_fe = d_st = None
    if tys.size:
        d_ys, d_xs = push(ctx, tys.astype(np.int64), txs.astype(np.int64))
        d_fe = torch.stack([res['Fpsf'][d_ys, d_xs], res['Fpsferr'][d_ys, d_xs]])
    if frame_stats:
        # statistics over the unmasked pixels (new frame's mask), clipped like zogy's header values
        d_st = torch.stack([frame_clipped_stats_enqueue(ctx, res['Scorr'], new_mask),
                            frame_clipped_stats_enqueue(ctx, res['Fpsferr'], new_mask)])
    back = [t for t in (d_fe, d_st) if t is not None]
    got = (fetch(ctx, *back) if len(back) > 1 else [fetch(ctx, back[0])]) if back else []
    fe = got.pop(0) if d_fe is not None else None
    st = got.pop(0) if d_st is not None else None
    if tys.size:
        res['transients'] = [dict(y=y, x=x, scorr=sc, fpsf=f, fpsferr=e)
                             for y, x, sc, f, e in zip(tys.tolist(), txs.tolist(), tsc.tolist(), fe[0].tolist(), fe[1].tolist())]
    else:
        res['transients'] = []
    hdr['Z-P'] = (True, 'successfully processed by ZOGY?')
    for h in (hdr, hdr_t):
        h['Z-SIZE'] = (size, '[pix] size of (square) ZOGY subimages')
        h['Z-BSIZE'] = (border, '[pix] size of ZOGY subimage borders')
    hdr_t['Z-DX'] = (float(np.median(dx)), '[pix] dx median offset full image')
    hdr_t['Z-DY'] = (float(np.median(dy)), '[pix] dy median offset full image')
    hdr_t['Z-FNR'] = (float(np.median(fratio)), 'median flux ratio (Fnew/Fref) full image')
    if frame_stats:
        hdr_t['Z-SCMED'] = (float(st[0, 1]), 'median Scorr full image')
        hdr_t['Z-SCSTD'] = (float(st[0, 3]), 'sigma (STD) Scorr full image')
        hdr_t['Z-FPEMED'] = (float(st[1, 1]), '[e-] median Fpsferr full image')
        hdr_t['Z-FPESTD'] = (float(st[1, 3]), '[e-] sigma (STD) Fpsferr full image')
    hdr_t['T-NSIGMA'] = (float(nsig), '[sigma] transient detection threshold')
    hdr_t['T-NTRANS'] = (ntrans, 'number of transient candidates')
    res['header'] = _HeaderView(hdr, hdr_t)
    res['header_new'], res['header_trans'] = hdr, hdr_t
    res['scal'] = scal
    return res


class _HeaderView(dict):
    """header_new additions as {KEY: (value, comment)}; item access also finds header_trans
    keys and returns plain values (what the callers of the tensor-level function read)"""

    def __init__(self, hdr, hdr_t):
        dict.__init__(self, hdr)
        self._t = hdr_t

    def __getitem__(self, k):
        v = dict.__getitem__(self, k) if dict.__contains__(self, k) else self._t[k]
        return v[0] if isinstance(v, tuple) else v
