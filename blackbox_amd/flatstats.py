"""get_flatstats (blackbox.py:3661-3820): header statistics of a reduced flat frame.

The reference draws an unseeded random 20 % (10 %) subsample for FLATMED/FLATSTD and the
sub-image medians (get_rand_indices), so those header values are estimators; here they are
evaluated over ALL valid pixels (the quantity the subsample estimates), which is
deterministic.  STATSEC and per-channel values are exact in the reference too.
One C-ABI call per family of rectangles (bbx_rect_stats): exact medians by bracketed select,
float64 moments."""
import ctypes as C

import numpy as np
import torch
from scipy import ndimage

from . import settings
from ._lib import lib, check
from . import reduce as R

get_par = settings.get_par


def _check_section(data, mask, y0, x0, ny, nx, ysz, xsz):
    """the library addresses the section through offset pointers: everything must lie
    inside the frame (an out-of-range section would be a device fault, not an error code)"""
    if data.dim() != 2 or data.dtype != torch.float32 or not data.is_contiguous():
        raise ValueError('contiguous 2-D float32 frame expected')
    NY, NX = data.shape
    if mask is not None and (mask.dtype != torch.uint8 or tuple(mask.shape) != (NY, NX) or not mask.is_contiguous()):
        raise ValueError('mask: contiguous uint8 of the frame shape expected')
    if not (0 <= y0 and 0 <= x0 and ny >= 1 and nx >= 1 and y0 + ny <= NY and x0 + nx <= NX):
        raise ValueError('section [{}:{}, {}:{}] outside the {}x{} frame'.format(y0, y0 + ny, x0, x0 + nx, NY, NX))
    if ysz < 1 or xsz < 1 or ny % ysz or nx % xsz or (ny // ysz) * (nx // xsz) > 64:
        raise ValueError('bad segmentation {}x{} of a {}x{} section'.format(ysz, xsz, ny, nx))


def rect_stats(ctx, data, mask, y0, x0, ny, nx, ysz, xsz):
    """bbx_rect_stats on the section [y0:y0+ny, x0:x0+nx] of the device frame [data] cut
    into segments of ysz x xsz -> numpy float64 [nseg, 8]
    (n, median, mean, sigma, n_low, sigma_low, 0, 0)"""
    NY, NX = data.shape
    _check_section(data, mask, y0, x0, ny, nx, ysz, xsz)
    nseg = (ny // ysz) * (nx // xsz)
    out = torch.empty((nseg, 8), dtype=torch.float64, device=data.device)
    off = y0 * NX + x0
    dptr = C.c_void_p(data.data_ptr() + 4 * off)
    mptr = C.c_void_p(mask.data_ptr() + off) if mask is not None else C.c_void_p(0)
    check(lib.bbx_rect_stats(ctx.h, ny, nx, NX, dptr, mptr, ysz, xsz, R._ptr(out), ctx.stream()), 'bbx_rect_stats', ctx.h)
    return out.cpu().numpy()


def rect_clipped_stats(ctx, data, mask, y0, x0, ny, nx, ysz, xsz, sigma=3.0, maxiters=5, skip_zero=True):
    """sigma_clipped_stats(..., mask_value=0) per segment -> numpy [nseg, 8] (n, median, mean, sigma, ...)"""
    NY, NX = data.shape
    _check_section(data, mask, y0, x0, ny, nx, ysz, xsz)
    nseg = (ny // ysz) * (nx // xsz)
    out = torch.empty((nseg, 8), dtype=torch.float64, device=data.device)
    off = y0 * NX + x0
    dptr = C.c_void_p(data.data_ptr() + 4 * off)
    mptr = C.c_void_p(mask.data_ptr() + off) if mask is not None else C.c_void_p(0)
    check(lib.bbx_rect_clipped_stats(ctx.h, ny, nx, NX, dptr, mptr, ysz, xsz, float(sigma), int(maxiters),
                                     1 if skip_zero else 0, R._ptr(out), ctx.stream()), 'bbx_rect_clipped_stats', ctx.h)
    return out.cpu().numpy()


def _finite(value, repl='None'):
    return value if np.isfinite(value) else repl


def _ratio(a, b):
    return a / b if (a != 'None' and b != 'None') else 'None'


def get_flatstats(ctx, data, header, data_mask, tel=None, statsec=None, subsize=None, ysize_chan=None,
                  xsize_chan=None):
    """data, data_mask: device tensors of the reduced flat.  Fills the header like the
    reference (same keywords, same comments)."""
    NY, NX = data.shape
    sec = statsec if statsec is not None else get_par(settings.flat_norm_sec, tel)
    header['STATSEC'] = ('[{}:{},{}:{}]'.format(sec[0].start + 1, sec[0].stop + 1, sec[1].start + 1, sec[1].stop + 1),
                         'pre-defined statistics section [y1:y2,x1:x2]')
    h, w = sec[0].stop - sec[0].start, sec[1].stop - sec[1].start
    st = rect_stats(ctx, data, data_mask, sec[0].start, sec[1].start, h, w, h, w)[0]
    med_sec, std_sec = _finite(np.float32(st[1])), _finite(np.float32(st[3]))
    header['MEDSEC'] = (med_sec, '[e-] median flat over STATSEC')
    header['STDSEC'] = (std_sec, '[e-] sigma (STD) flat over STATSEC')
    header['RSTDSEC'] = (_ratio(std_sec, med_sec), 'relative sigma (STD) flat over STATSEC')

    st = rect_stats(ctx, data, data_mask, 0, 0, NY, NX, NY, NX)[0]
    med, std = _finite(np.float32(st[1])), _finite(np.float32(st[3]))
    header['FLATMED'] = (med, '[e-] median flat')
    header['FLATSTD'] = (std, '[e-] sigma (STD) flat')
    header['FLATRSTD'] = (_ratio(std, med), 'relative sigma (STD) flat')

    ysz = ysize_chan or NY // 2
    xsz = xsize_chan or NX // 8
    st = rect_stats(ctx, data, None, 0, 0, NY, NX, ysz, xsz)
    for c in range(16):
        m, s = _finite(np.float32(st[c, 1])), _finite(np.float32(st[c, 3]))
        header['FLATM{}'.format(c + 1)] = (m, '[e-] channel {} median flat (bias-subtracted)'.format(c + 1))
        header['FLATS{}'.format(c + 1)] = (s, '[e-] channel {} sigma (STD) flat'.format(c + 1))
        header['FLATRS{}'.format(c + 1)] = (_ratio(s, m), 'channel {} relative sigma (STD) flat'.format(c + 1))

    sub = subsize or settings.subimage_size
    ns = NY // sub
    if ns < 3 or ns * sub > NX:
        raise ValueError('frame {}x{} too small for {}-pixel sub-image statistics'.format(NY, NX, sub))
    st = rect_stats(ctx, data, data_mask, 0, 0, ns * sub, ns * sub, sub, sub).reshape(ns, ns, 8)
    mini_median, mini_std = st[:, :, 1], st[:, :, 5]
    mask_cntr = ndimage.binary_erosion(np.ones(mini_median.shape, dtype=bool))
    with np.errstate(invalid='ignore', divide='ignore'):
        minimum, maximum = np.amin(mini_median[mask_cntr]), np.amax(mini_median[mask_cntr])
        danstat = _finite(np.abs((maximum - minimum) / (maximum + minimum)))
    header['NSUBSTOT'] = (mask_cntr.size, 'number of subimages available for statistics')
    header['NSUBS'] = (int(np.sum(mask_cntr)), 'number of subimages used for statistics')
    header['RDIF-MAX'] = (danstat, '(max(subs)-min(subs)) / (max(subs)+min(subs))')
    nz = mini_median[mask_cntr] != 0
    rstd_max = np.amax(mini_std[mask_cntr][nz] / np.abs(mini_median[mask_cntr][nz])) if np.sum(nz) != 0 else 'None'
    header['RSTD-MAX'] = (rstd_max, 'max. relative sigma (STD) of subimages')
    return header
