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
from ._lib import lib, check

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


def zoom_plan(mini, box, channels=None):
    """host part of mini2back: B-spline coefficients of the (edge-padded) mini image and the
    per-row / per-column tap tables for scipy.ndimage.zoom(mini, box, order=3, mode='nearest');
    channels=(cy, cx) boxes -> every channel block gets its own padded coefficient patch
    (interp_Xchan=False)."""
    mini = np.asarray(mini, np.float64)
    nby, nbx = mini.shape
    cy, cx = (nby, nbx) if channels is None else channels
    py, px = cy + 2 * NPAD, cx + 2 * NPAD
    coef = np.empty(((nby // cy) * py, (nbx // cx) * px))
    fy = np.empty(nby * box, np.int32); wy = np.empty((nby * box, 4))
    fx = np.empty(nbx * box, np.int32); wx = np.empty((nbx * box, 4))
    for iy in range(nby // cy):
        f, w = _axis_map(cy, cy * box, iy * py)
        fy[iy * cy * box:(iy + 1) * cy * box], wy[iy * cy * box:(iy + 1) * cy * box] = f, w
    for ix in range(nbx // cx):
        f, w = _axis_map(cx, cx * box, ix * px)
        fx[ix * cx * box:(ix + 1) * cx * box], wx[ix * cx * box:(ix + 1) * cx * box] = f, w
    for iy in range(nby // cy):
        for ix in range(nbx // cx):
            blk = np.pad(mini[iy * cy:(iy + 1) * cy, ix * cx:(ix + 1) * cx], NPAD, mode='edge')
            coef[iy * py:(iy + 1) * py, ix * px:(ix + 1) * px] = ndimage.spline_filter(blk, order=3, mode='nearest',
                                                                                      output=np.float64)
    return coef, fy, wy, fx, wx


def mini2back(ctx, mini, shape, bkg_boxsize=None, interp_Xchan=True, subtract_from=None, want_bkg=True):
    """zogy.mini2back(data_mini, data_shape, order_interp=3, bkg_boxsize, interp_Xchan): full-
    frame background from the mini image; with subtract_from the same pass does `data -= bkg`."""
    box = bkg_boxsize or settings.bkg_boxsize
    mini_h = mini.cpu().numpy() if torch.is_tensor(mini) else np.asarray(mini)
    channels = None
    if not interp_Xchan:
        channels = (mini_h.shape[0] // settings.ny, mini_h.shape[1] // settings.nx)
    coef, fy, wy, fx, wx = zoom_plan(mini_h, box, channels)
    dev = ctx.device
    d_coef = torch.from_numpy(coef).to(dev)
    d_fy, d_wy = torch.from_numpy(fy).to(dev), torch.from_numpy(wy).to(dev)
    d_fx, d_wx = torch.from_numpy(fx).to(dev), torch.from_numpy(wx).to(dev)
    ny, nx = shape
    bkg = torch.empty((ny, nx), dtype=torch.float32, device=dev) if want_bkg else None
    check(lib.bbx_spline_zoom(ctx.h, ny, nx, _p(d_coef), coef.shape[0], coef.shape[1], _p(d_fy), _p(d_wy), _p(d_fx),
                              _p(d_wx), _p(subtract_from), _p(bkg), ctx.stream()), 'bbx_spline_zoom', ctx.h)
    return bkg


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


def psf_optflux(ctx, D, V, psfs, ys, xs):
    """zogy.get_psfoptflux at integer positions -> (flux, fluxerr) float32 device tensors"""
    nsrc, S, _ = psfs.shape
    dev = ctx.device
    d_ys = torch.as_tensor(np.asarray(ys, np.int32)).to(dev)
    d_xs = torch.as_tensor(np.asarray(xs, np.int32)).to(dev)
    flux = torch.empty(nsrc, dtype=torch.float32, device=dev)
    err = torch.empty(nsrc, dtype=torch.float32, device=dev)
    ny, nx = D.shape
    check(lib.bbx_psf_optflux(ctx.h, ny, nx, _p(D), _p(V), _p(psfs), S, nsrc, _p(d_ys), _p(d_xs), _p(flux), _p(err),
                              ctx.stream()), 'bbx_psf_optflux', ctx.h)
    return flux, err
