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
