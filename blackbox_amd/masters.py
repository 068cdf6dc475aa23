"""Master calibration frames on the GPU (reference master_prep, blackbox.py:4625-5247).

Only the bulk array work of master_prep is here: normalisation of the individual flats by
their MEDSEC, the pixel-wise median of the cube and the edge / non-positive fix of the
master flat (4929-4941, 4984, 5071-5073).  Frame selection by date, header bookkeeping
and the GAINCF / MBMEAN header statistics are orchestration and not part of this round.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, check


def master_median(ctx, frames, imgtype, medsec=None, bpm=None):
    """frames: list of contiguous float32 device tensors of equal shape (the reduced
    bias/dark/flat frames).  imgtype 'flat': medsec = list of the frames' MEDSEC header
    values (median over set_bb.flat_norm_sec); bpm: uint8 bad-pixel mask or None.
    -> master float32 device tensor"""
    n = len(frames)
    if n < 1 or n > 32:
        raise ValueError('1..32 frames expected, got {}'.format(n))
    shape = frames[0].shape
    for f in frames:
        if f.dtype != torch.float32 or f.shape != shape or not f.is_contiguous():
            raise ValueError('frames must be contiguous float32 tensors of equal shape')
    out = torch.empty(shape, dtype=torch.float32, device=ctx.device)
    ptrs = (C.c_void_p * n)(*[f.data_ptr() for f in frames])
    norm = None
    if imgtype == 'flat':
        if medsec is None or len(medsec) != n:
            raise ValueError('flat frames need their MEDSEC values')
        norm = (C.c_float * n)(*[float(m) for m in medsec])
    check(lib.bbx_median_stack(ctx.h, out.numel(), n, ptrs, norm,
                               C.c_void_p(bpm.data_ptr()) if (bpm is not None and imgtype == 'flat') else None,
                               1 if (imgtype == 'flat' and bpm is not None) else 0,
                               C.c_void_p(out.data_ptr()), ctx.stream()), 'bbx_median_stack', ctx.h)
    return out


def _rect_scale(ctx, t, y0, x0, ny, nx, factor, divide):
    NY, NX = t.shape
    if not (0 <= y0 and 0 <= x0 and ny >= 1 and nx >= 1 and y0 + ny <= NY and x0 + nx <= NX):
        raise ValueError('section outside the frame')
    check(lib.bbx_rect_scale(ctx.h, ny, nx, NX, C.c_void_p(t.data_ptr() + 4 * (y0 * NX + x0)), C.c_float(float(factor)),
                             1 if divide else 0, ctx.stream()), 'bbx_rect_scale', ctx.h)


def gain_correction_factors(ctx, master, header, ysize_chan=None, xsize_chan=None, nrows_v=200, nrows_h=2000,
                            ncols=200):
    """GAINCF{c} of a master flat (blackbox.py:5085-5161): match the channels vertically
    with the medians of the [nrows_v] rows next to the central row boundary, then
    horizontally, pair of channel columns by pair, with the medians of [ncols] columns
    either side of each boundary over 2*[nrows_h] central rows; factors normalised to a
    mean of one.  Medians are exact (bbx_rect_stats); the frame copy is scaled channel by
    channel on the device (float32, like numpy's in-place ops on the float32 master)."""
    import numpy as np
    from . import flatstats
    NY, NX = master.shape
    ysz, xsz = ysize_chan or NY // 2, xsize_chan or NX // 8
    def pyslice(start, stop, n):
        """[start:stop] of an axis of length n with Python's rules (the reference slices numpy arrays
        with these numbers; strips larger than the frame are clamped, a negative start wraps once)"""
        a, b, _ = slice(start, stop).indices(n)
        if b <= a:
            raise ValueError('empty statistics strip [{}:{}] on an axis of {}'.format(start, stop, n))
        return a, b - a
    corr = master.clone()
    med = np.zeros(16)
    for c in range(16):
        iy, ix = divmod(c, 8)
        ya, nr = pyslice(-nrows_v, None, ysz) if iy == 0 else pyslice(0, nrows_v, ysz)      # data_chan[-nrows:, :] / [0:nrows, :]
        med[c] = np.float32(flatstats.rect_stats(ctx, corr, None, iy * ysz + ya, ix * xsz, nr, xsz, nr, xsz)[0, 1])
        _rect_scale(ctx, corr, iy * ysz, ix * xsz, ysz, xsz, med[c], True)
    factor = 1.0 / med
    for i in range(1, 8):
        x_index = i * xsz
        ya, nr = pyslice(ysz - nrows_h, ysz + nrows_h, NY)
        xa, nc1 = pyslice(x_index - ncols, x_index, NX)
        xb, nc2 = pyslice(x_index, x_index + ncols, NX)
        m1 = np.float32(flatstats.rect_stats(ctx, corr, None, ya, xa, nr, nc1, nr, nc1)[0, 1])
        m2 = np.float32(flatstats.rect_stats(ctx, corr, None, ya, xb, nr, nc2, nr, nc2)[0, 1])
        ratio = np.float32(m1) / np.float32(m2)
        _rect_scale(ctx, corr, 0, i * xsz, ysz, xsz, ratio, False)
        _rect_scale(ctx, corr, ysz, i * xsz, ysz, xsz, ratio, False)
        factor[i] *= ratio
        factor[i + 8] *= ratio
    factor /= np.mean(factor)
    for c in range(16):
        header['GAINCF{}'.format(c + 1)] = (float(factor[c]), 'channel {} gain correction factor'.format(c + 1))
    return factor


def master_level_stats(ctx, master, header, imgtype, ysize_chan=None, xsize_chan=None):
    """header statistics of a master bias / dark (blackbox.py:5167-5230): sigma-clipped mean
    and sigma of the frame and of each channel, zeros masked.  Keywords MBMEAN MBRDN
    MBIASM{c} MBRDN{c} (bias) or MDMEAN MDRDN MDARKM{c} MDRDN{c} (dark)."""
    from . import flatstats
    NY, NX = master.shape
    ysz, xsz = ysize_chan or NY // 2, xsize_chan or NX // 8
    full = flatstats.rect_clipped_stats(ctx, master, None, 0, 0, NY, NX, NY, NX)[0]
    chan = flatstats.rect_clipped_stats(ctx, master, None, 0, 0, NY, NX, ysz, xsz)
    if imgtype == 'bias':
        k = ('MBMEAN', 'MBRDN', 'MBIASM{}', 'MBRDN{}', 'bias')
    elif imgtype == 'dark':
        k = ('MDMEAN', 'MDRDN', 'MDARKM{}', 'MDRDN{}', 'dark')
    else:
        raise ValueError('bias or dark expected')
    header[k[0]] = (float(full[2]), '[e-] mean master {}'.format(k[4]))
    header[k[1]] = (float(full[3]), '[e-] sigma (STD) master {}'.format(k[4]))
    for c in range(16):
        header[k[2].format(c + 1)] = (float(chan[c, 2]), '[e-] channel {} mean master {}'.format(c + 1, k[4]))
    for c in range(16):
        header[k[3].format(c + 1)] = (float(chan[c, 3]), '[e-] channel {} sigma (STD) master {}'.format(c + 1, k[4]))
    return full, chan
