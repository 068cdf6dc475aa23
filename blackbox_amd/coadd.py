"""blackbox_amd.coadd -- reference co-add on the GPU.

Host-side mirror of the data path of the reference's buildref.py:

  prep_inputimage   the array part of prep_inputimages (buildref.py:2442-2777): weights from the
                    background-sigma image and the mask, background subtraction, edge pixels
  scale_chan_zps    buildref.py:3019-3047
  resample          SWarp's -RESAMPLING_TYPE LANCZOS3 step (buildref.py:1748)
  combine           SWarp's -COMBINE_TYPE step (buildref.py:1733, 1815; CLIPPED 1780-1788)
  clipped2mask      clipped2mask_loop + pass_filters (buildref.py:3686-3873): clip log -> input-frame masks
  imcombine         resample every prepared image onto the output frame and combine them: the
                    data path of imcombine_mp (buildref.py:1425-2000) between reading the inputs
                    and writing the co-add

Everything runs through the HIP library (bbx_coadd_prep / bbx_resample_lanczos3 /
bbx_coadd_combine / bbx_rect_scale); there is no CPU fallback.  The sky <-> pixel mapping is the
gnomonic (TAN) projection with a CD matrix, evaluated on the host (float64, numpy) on a coarse
lattice of output pixels -- what SWarp does too -- and interpolated on the device.  Distortion
terms (PV / SIP) are not handled: such a header has to be resolved into the lattice by the caller
(`grid=`).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, settings
from ._lib import lib, check
from .reduce import _ptr, _expect

COMBINE_TYPES = {'weighted': 0, 'average': 1, 'median': 2, 'clipped': 3, 'min': 4, 'max': 5, 'sum': 6}
GRID_STEP = 32


# ---- prep_inputimages ----------------------------------------------------------------------
def discard_bits(masktype_discard, mask_value=None):
    """the union of the mask types the reference loops over (buildref.py:2611-2616)"""
    mask_value = mask_value or settings.mask_value
    bits = 0
    for val in mask_value.values():
        if masktype_discard & val != 0:
            bits |= val
    return bits


def prep_inputimage(ctx, data, bkg, bkg_std, mask, masktype_discard=0, nimages=2, edge_value=None):
    """in place on [data]; -> (data, weights).  bkg may be None (BKG-SUB image)."""
    ny, nx = data.shape
    _expect(data, torch.float32, (ny, nx), 'data')
    _expect(bkg_std, torch.float32, (ny, nx), 'bkg_std')
    _expect(mask, torch.uint8, (ny, nx), 'mask')
    if bkg is not None:
        _expect(bkg, torch.float32, (ny, nx), 'bkg')
    if edge_value is None:
        edge_value = settings.mask_value['edge']
    w = torch.empty_like(data)
    bits = discard_bits(masktype_discard) if nimages > 1 else 0
    check(lib.bbx_coadd_prep(ctx.h, data.numel(), _ptr(data), _ptr(bkg), _ptr(bkg_std), _ptr(mask), int(bits),
                             int(edge_value), _ptr(w), ctx.stream()), 'bbx_coadd_prep', ctx.h)
    return data, w


def scale_chan_zps(ctx, data, header, geom):
    """data[channel c] *= 10**(0.4*(PC-ZP - PC-ZP{c})), header PC-ZP{c} <- PC-ZP"""
    zp = header['PC-ZP']
    ysz, xsz = geom.ysize_chan, geom.xsize_chan
    for c in range(16):
        key = 'PC-ZP{}'.format(c + 1)
        if key not in header:
            continue
        factor = 10 ** (0.4 * (zp - header[key]))
        iy, ix = c // 8, c % 8
        sub = data[iy * ysz:(iy + 1) * ysz, ix * xsz:(ix + 1) * xsz]
        check(lib.bbx_rect_scale(ctx.h, ysz, xsz, data.shape[1], _ptr(sub), float(np.float32(factor)), 0, ctx.stream()),
              'bbx_rect_scale', ctx.h)
        header[key] = zp


# ---- sky <-> pixel (TAN + CD) -----------------------------------------------------------------
class TanWCS:
    """gnomonic projection: CRVAL (deg), CRPIX (1-based FITS pixels), CD (deg/pixel)"""

    def __init__(self, crval, crpix, cd):
        self.crval = np.asarray(crval, np.float64)
        self.crpix = np.asarray(crpix, np.float64)
        self.cd = np.asarray(cd, np.float64).reshape(2, 2)
        self.cdi = np.linalg.inv(self.cd)

    @classmethod
    def from_header(cls, h):
        if 'CD1_1' in h:
            cd = [[h['CD1_1'], h.get('CD1_2', 0.0)], [h.get('CD2_1', 0.0), h['CD2_2']]]
        else:
            cd = [[h['CDELT1'] * h.get('PC1_1', 1.0), h['CDELT1'] * h.get('PC1_2', 0.0)],
                  [h['CDELT2'] * h.get('PC2_1', 0.0), h['CDELT2'] * h.get('PC2_2', 1.0)]]
        return cls([h['CRVAL1'], h['CRVAL2']], [h['CRPIX1'], h['CRPIX2']], cd)

    def pix2sky(self, x, y):
        """0-based pixel centres -> (ra, dec) in degrees"""
        u = x + 1.0 - self.crpix[0]; v = y + 1.0 - self.crpix[1]
        xi = np.deg2rad(self.cd[0, 0] * u + self.cd[0, 1] * v)
        eta = np.deg2rad(self.cd[1, 0] * u + self.cd[1, 1] * v)
        a0, d0 = np.deg2rad(self.crval)
        den = np.cos(d0) - eta * np.sin(d0)
        ra = a0 + np.arctan2(xi, den)
        dec = np.arctan2((np.sin(d0) + eta * np.cos(d0)) * np.cos(ra - a0), den)
        return np.rad2deg(ra), np.rad2deg(dec)

    def sky2pix(self, ra, dec):
        a, d = np.deg2rad(ra), np.deg2rad(dec)
        a0, d0 = np.deg2rad(self.crval)
        cosc = np.sin(d0) * np.sin(d) + np.cos(d0) * np.cos(d) * np.cos(a - a0)
        xi = np.rad2deg(np.cos(d) * np.sin(a - a0) / cosc)
        eta = np.rad2deg((np.cos(d0) * np.sin(d) - np.sin(d0) * np.cos(d) * np.cos(a - a0)) / cosc)
        u = self.cdi[0, 0] * xi + self.cdi[0, 1] * eta
        v = self.cdi[1, 0] * xi + self.cdi[1, 1] * eta
        return u + self.crpix[0] - 1.0, v + self.crpix[1] - 1.0


def projection_grid(wcs_in, wcs_out, out_shape, step=GRID_STEP):
    """input pixel coordinates of the output lattice nodes -> float64 [gny][gnx][2] (x, y)"""
    out_ny, out_nx = out_shape
    gy = np.arange(0, out_ny + step, step, dtype=np.float64)
    gx = np.arange(0, out_nx + step, step, dtype=np.float64)
    yy, xx = np.meshgrid(gy, gx, indexing='ij')
    ra, dec = wcs_out.pix2sky(xx, yy)
    xin, yin = wcs_in.sky2pix(ra, dec)
    return np.ascontiguousarray(np.stack([xin, yin], axis=-1))


# ---- resampling and combination ------------------------------------------------------------
def resample(ctx, img, wimg, grid, out_shape, fscale=1.0, step=GRID_STEP, out=None, wout=None):
    """-> (resampled float32 [out_shape], weights).  grid: numpy float64 [gny][gnx][2] or a
    device tensor of that shape"""
    in_ny, in_nx = img.shape
    _expect(img, torch.float32, (in_ny, in_nx), 'image')
    _expect(wimg, torch.float32, (in_ny, in_nx), 'weights')
    out_ny, out_nx = out_shape
    if not torch.is_tensor(grid):
        grid = torch.from_numpy(np.ascontiguousarray(grid, np.float64)).to(ctx.device)
    gny, gnx = int(grid.shape[0]), int(grid.shape[1])
    _expect(grid, torch.float64, (gny, gnx, 2), 'grid')
    if (out_ny - 1) // step + 1 >= gny or (out_nx - 1) // step + 1 >= gnx:
        raise ValueError('projection grid {}x{} does not cover a {}x{} frame at step {}'.format(gny, gnx, out_ny, out_nx, step))
    if out is None:
        out = torch.empty(out_shape, dtype=torch.float32, device=ctx.device)
    if wout is None:
        wout = torch.empty(out_shape, dtype=torch.float32, device=ctx.device)
    _expect(out, torch.float32, tuple(out_shape), 'out')
    _expect(wout, torch.float32, tuple(out_shape), 'wout')
    check(lib.bbx_resample_lanczos3(ctx.h, in_ny, in_nx, _ptr(img), _ptr(wimg), out_ny, out_nx, _ptr(grid), gny, gnx,
                                    int(step), float(fscale), _ptr(out), _ptr(wout), ctx.stream()),
          'bbx_resample_lanczos3', ctx.h)
    return out, wout


def combine(ctx, cube, wcube, combine_type='weighted', nsigma_clip=4.0, A_swarp=0.3, clipmask=False):
    """cube, wcube: float32 device tensors [n][ny][nx] -> (out, wout, nclip int64[n] tensor,
    clip mask uint8 [n][ny][nx] or None)"""
    t = combine_type.lower()
    if t not in COMBINE_TYPES:
        raise ValueError('[combine_type] method "{}" should be one of {}'.format(combine_type, sorted(COMBINE_TYPES)))
    n, ny, nx = cube.shape
    if n > 32:
        raise ValueError('at most 32 images per combination, got {}'.format(n))
    _expect(cube, torch.float32, (n, ny, nx), 'cube')
    _expect(wcube, torch.float32, (n, ny, nx), 'wcube')
    out = torch.empty((ny, nx), dtype=torch.float32, device=ctx.device)
    wout = torch.empty((ny, nx), dtype=torch.float32, device=ctx.device)
    nclip = torch.zeros(n, dtype=torch.int64, device=ctx.device)
    cm = torch.empty((n, ny, nx), dtype=torch.uint8, device=ctx.device) if (clipmask and t == 'clipped') else None
    ns = torch.empty((n, ny, nx), dtype=torch.float32, device=ctx.device) if cm is not None else None
    check(lib.bbx_coadd_combine(ctx.h, n, ny * nx, _ptr(cube), _ptr(wcube), ny * nx, COMBINE_TYPES[t],
                                float(nsigma_clip), float(A_swarp), _ptr(out), _ptr(wout), _ptr(cm), _ptr(ns), _ptr(nclip),
                                ctx.stream()), 'bbx_coadd_combine', ctx.h)
    if cm is not None:
        return out, wout, nclip, (cm, ns)
    return out, wout, nclip, None


def clipped2mask(ctx, clip, nsig, grid, in_shape, data_mask, weights, nsigma_clip, fwhm, step=GRID_STEP,
                 sat_bits=None, fsize=(5, 1), fmax=(4, 1)):
    """clipped2mask_loop (buildref.py:3686-3783) for one input image: clip / nsig = that image's
    planes of the clip log (combine(..., 'clipped', clipmask=True)), grid = its projection lattice;
    zeroes [weights] (in place) where the filtered clipped pixels fall -> (mask uint8, number)"""
    out_ny, out_nx = clip.shape
    in_ny, in_nx = in_shape
    _expect(clip, torch.uint8, (out_ny, out_nx), 'clip')
    _expect(nsig, torch.float32, (out_ny, out_nx), 'nsig')
    _expect(data_mask, torch.uint8, (in_ny, in_nx), 'data_mask')
    _expect(weights, torch.float32, (in_ny, in_nx), 'weights')
    if not torch.is_tensor(grid):
        grid = torch.from_numpy(np.ascontiguousarray(grid, np.float64)).to(ctx.device)
    gny, gnx = int(grid.shape[0]), int(grid.shape[1])
    _expect(grid, torch.float64, (gny, gnx, 2), 'grid')
    if sat_bits is None:
        sat_bits = 0
        for key, val in settings.mask_value.items():
            if 'saturated' in key:
                sat_bits |= val
    nf = len(fsize)
    fsigma = [float(nsigma_clip), 4.0][:nf] if nf <= 2 else [float(nsigma_clip)] + [4.0] * (nf - 1)
    mask_im = torch.empty((in_ny, in_nx), dtype=torch.uint8, device=ctx.device)
    nmasked = torch.zeros(1, dtype=torch.int64, device=ctx.device)
    check(lib.bbx_clipped2mask(ctx.h, out_ny, out_nx, _ptr(clip), _ptr(nsig), _ptr(grid), gny, gnx, int(step), in_ny, in_nx,
                               _ptr(data_mask), int(sat_bits), float((5 * fwhm) ** 2), nf, (C.c_int * nf)(*[int(v) for v in fsize]),
                               (C.c_float * nf)(*fsigma), (C.c_int * nf)(*[int(v) for v in fmax]), _ptr(weights), _ptr(mask_im),
                               _ptr(nmasked), ctx.stream()), 'bbx_clipped2mask', ctx.h)
    return mask_im, nmasked


def imcombine(ctx, images, weights, wcs_list, wcs_out, out_shape, combine_type='weighted', fscale=None,
              nsigma_clip=4.0, A_swarp=0.3, step=GRID_STEP, clipmask=False, masks=None, fwhm=None):
    """images / weights: prepared device tensors (prep_inputimage); wcs_list: TanWCS per image.
    -> (co-add, weights, nclip, clip log).  With combine_type 'clipped' and masks + fwhm given, the
    reference's two passes run (buildref.py:1773-1833): CLIPPED with the clip log, clipped2mask on
    every input image (its weights are changed in place), then WEIGHTED."""
    n = len(images)
    fscale = [1.0] * n if fscale is None else list(fscale)
    cube = torch.empty((n,) + tuple(out_shape), dtype=torch.float32, device=ctx.device)
    wcube = torch.empty_like(cube)
    grids = [torch.from_numpy(projection_grid(wcs_list[k], wcs_out, out_shape, step)).to(ctx.device) for k in range(n)]
    for k in range(n):
        resample(ctx, images[k], weights[k], grids[k], out_shape, fscale[k], step, out=cube[k], wout=wcube[k])
    two_pass = combine_type.lower() == 'clipped' and masks is not None and fwhm is not None
    res = combine(ctx, cube, wcube, combine_type, nsigma_clip, A_swarp, clipmask or two_pass)
    if not two_pass:
        return res
    cm, ns = res[3]
    for k in range(n):
        clipped2mask(ctx, cm[k], ns[k], grids[k], tuple(images[k].shape), masks[k], weights[k], nsigma_clip, fwhm[k], step)
        resample(ctx, images[k], weights[k], grids[k], out_shape, fscale[k], step, out=cube[k], wout=wcube[k])
    out, wout, _, _ = combine(ctx, cube, wcube, 'weighted')
    return out, wout, res[2], res[3]
