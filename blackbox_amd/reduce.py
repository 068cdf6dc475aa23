"""Per-image reduction on one MI355X: Python mirror of the reference's stage
functions (same names, same argument meaning, same header keywords) on top of
the HIP C ABI (include/bbx.h).  Order of operations = blackbox_reduce,
blackbox.py:1451-1990.

Arrays live in HBM as torch tensors (torch is used for device memory and streams
only); every stage calls hand-written kernels through ctypes.  There is no CPU
fallback: importing this module without libbbx_hip.so raises.

Fusion map (reference statement -> kernel):
  gain_corr 7460 + os_corr 6553/6844-6847 + `data -= mbias` 1679 + mask_init
  4408-4414/4494-4498/4538 + `data /= mflat` 1825      -> bbx_calibrate (one pass)
  os_corr strip statistics 6480-6490, 6572-6573, 6636-6640 -> bbx_overscan_stats,
                                                              bbx_vos_std, bbx_satcol_counts
  mask_init 4504-4562 + fill_sat_holes 4584-4596       -> bbx_mask_finish
  cosmics_corr 4259-4370 (astroscrappy)                 -> bbx_lacosmic
  xtalk_corr 7138-7258                                  -> bbx_xtalk
  mask_header 4601-4620, edge fill 1968-1974            -> bbx_mask_counts, bbx_edge_fill
"""
import ctypes as C

import numpy as np
import torch

import os

from . import _lib, overscan, settings
from ._lib import lib, check, Geom, BBX_RAW_U16, BBX_RAW_F32

get_par = settings.get_par
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Context:
    """one per worker process / GPU (bbx_ctx)"""

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError('blackbox_amd needs a GPU (MI355X); no CPU fallback')
        self.device = torch.device('cuda', device)
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        check(lib.bbx_ctx_create(device, C.byref(h)), 'bbx_ctx_create')
        self.h = h
        for kv in filter(None, os.environ.get('BBX_DEBUG_OPTIONS', '').split(',')):      # (tuning runs: "id=value,..." of include/bbx.h's BBX_OPT_*)
            k, v = kv.split('=')
            check(lib.bbx_set_option(self.h, int(k), int(v)), 'bbx_set_option', self.h)

    def stream(self):
        # (the raw handle of torch's current stream: ~0.3 us; torch.cuda.current_stream() builds a Stream object, ~5 us, and a
        # frame asks ~50 times)
        return C.c_void_p(_raw_stream(self.device.index))

    def set_lac_level_feed(self, on):
        """BBX_OPT_LAC_LEVEL_FEED (include/bbx.h): prepare LA-Cosmic's background level during the
        dense pass (True) or select it over the frame only when a frame needs it (False, default)"""
        check(lib.bbx_set_option(self.h, 1, 1 if on else 0), 'bbx_set_option', self.h)

    def sync(self):
        check(lib.bbx_sync(self.h, self.stream()), 'bbx_sync', self.h)

    def close(self):
        if self.h:
            lib.bbx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _expect(t, dtype, shape, name):
    """device pointers cross the C ABI without their extents: a wrong shape or dtype would be
    a device fault, so it is refused here"""
    if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous() or not t.is_cuda:
        raise ValueError('{}: expected contiguous {} device tensor of shape {}, got {} {}'.format(
            name, dtype, tuple(shape), t.dtype, tuple(t.shape)))


def raw_type_of(raw):
    if raw.dtype == torch.uint16:
        return BBX_RAW_U16
    if raw.dtype == torch.int16:
        raise TypeError('raw int16: apply BZERO and pass uint16')
    if raw.dtype == torch.float32:
        return BBX_RAW_F32
    raise TypeError('raw frame must be uint16 or float32, got {}'.format(raw.dtype))


def geometry(raw_shape, ysize_chan=None, xsize_chan=None):
    ysize_chan = ysize_chan or settings.ysize_chan
    xsize_chan = xsize_chan or settings.xsize_chan
    return Geom(int(raw_shape[0]), int(raw_shape[1]), int(ysize_chan), int(xsize_chan))


def define_sections(data_shape, xbin=1, ybin=1, tel=None, ysize_chan=None, xsize_chan=None):
    """blackbox.py:6334-6402 (host index helper; the kernels carry the same maths)"""
    ysize, xsize = data_shape
    ny, nx = settings.ny, settings.nx
    dy, dx = ysize // ny, xsize // nx
    ysc = (ysize_chan or settings.ysize_chan) // ybin
    xsc = (xsize_chan or settings.xsize_chan) // xbin
    ysize_os = (ysize - ny * ysc) // ny
    xsize_os = (xsize - nx * xsc) // nx
    chan_sec = tuple((slice(y, y + dy), slice(x, x + dx))
                     for y in range(0, ysize, dy) for x in range(0, xsize, dx))
    data_sec = tuple((slice(y, y + ysc), slice(x, x + xsc))
                     for y in range(0, ysize, dy + ysize_os) for x in range(0, xsize, dx))
    ncut_vert = max(5 // xbin, 1)
    os_sec_vert = tuple((slice(y, y + dy), slice(x + xsc + ncut_vert, x + dx - 1))
                        for y in range(0, ysize, dy) for x in range(0, xsize, dx))
    cut = ysize_os - max(10 // ybin, 1)
    os_sec_hori = tuple((slice(y, y + cut), slice(x, x + dx))
                        for y in range(dy - cut, dy + cut, cut) for x in range(0, xsize, dx))
    data_sec_red = tuple((slice(y, y + ysc), slice(x, x + xsc))
                         for y in range(0, ysize - ny * ysize_os, ysc)
                         for x in range(0, xsize - nx * xsize_os, xsc))
    return chan_sec, data_sec, os_sec_hori, os_sec_vert, data_sec_red


class OverscanSolution:
    """what os_corr determines before any pixel is rewritten"""
    __slots__ = ('vfit', 'oscan', 'd_vfit', 'd_oscan', 'header', 'aux')


def gain_corr(header, tel):
    """blackbox.py:7442-7465 (header part; the multiplication itself is fused into
    every kernel that reads the raw frame: float32 raw * float32(gain))"""
    gain = get_par(settings.gain, tel)
    for c in range(16):
        header['GAIN{}'.format(c + 1)] = (gain[c], '[e-/ADU] gain applied to channel {}'.format(c + 1))
    return gain


_FIT_THREADS = []


def _fit_threads():
    """a few host threads for the per-channel overscan fits of the serial path (created at first use)"""
    if not _FIT_THREADS:
        from concurrent.futures import ThreadPoolExecutor
        import os
        _FIT_THREADS.append(ThreadPoolExecutor(max_workers=max(2, min(8, (os.cpu_count() or 4)))))
    return _FIT_THREADS[0]


def _vos_header(header, c, coeffs, ok):
    for k, v in enumerate(coeffs):
        header['BIAS{}A{}'.format(c + 1, k)] = (float(v) if np.isfinite(v) else 'None',
                                               '[e-] channel {} vert. overscan A{} polyfit coeff'.format(c + 1, k))
    header['VFITOK{}'.format(c + 1)] = (bool(ok), 'channel {} vert. overscan polyfit finite?'.format(c + 1))


def _os_solve_stepwise(ctx, raw, header, tel, geom, data_limit, accum, mean_vos_col, hos, vfit, mean_vos, oscan, aux, d_std, g32, rt):
    """the channels one after the other, with the device step BlackGEM needs in between (saturated-column counts after
    the vertical fits) -> (failure or None, d_vfit)"""
    dev = ctx.device
    ny_raw, nx_raw = raw.shape
    ysz, xsz = geom.ysize_chan, geom.xsize_chan
    dy, dx = ny_raw // 2, nx_raw // 8
    hos_rows = dy - ysz - 10
    gain = get_par(settings.gain, tel)
    for c in range(16):
        fit, coeffs, ok, level = overscan.vos_polyfit(mean_vos_col[c], ysz, c, settings.voscan_poldeg)
        _vos_header(header, c, coeffs, ok)
        vfit[c] = fit
        mean_vos[c] = level
    d_vfit = torch.from_numpy(vfit.reshape(-1)).to(dev)
    mask_sat_rows = None
    if tel != 'ML1':
        lim = settings.os_ypix_lim[tel]
        satl = np.array(get_par(settings.satlevel, tel)) * np.array(gain)
        thr = _lib.f32x16(np.float32(0.9 * satl))
        d_cnt = torch.empty((2, 16, xsz), dtype=torch.int32, device=dev)
        check(lib.bbx_satcol_counts(ctx.h, C.byref(geom), _ptr(raw), rt, g32, _ptr(d_vfit), thr,
                                    int(lim[0]), int(lim[1]), _ptr(d_cnt), ctx.stream()),
              'bbx_satcol_counts', ctx.h)
        cnt = d_cnt.cpu().numpy()
        mask_sat_rows = (cnt[0] >= 3) | (cnt[1] >= 10)
    strips = []
    failure = None
    for c in range(16):
        # horizontal overscan rows after the vertical fit: float32 - float64 -> float32
        rl0 = (dy - hos_rows) if c < 8 else 0
        strip = (hos[c].astype(np.float64) - vfit[c][rl0:rl0 + hos_rows, None]).astype(np.float32)
        window = strip[:, xsz - 300:xsz]                   # blackbox.py:6565-6566: python slice on the dx-wide strip
        dlevel = overscan.clipped_stats_flat(window, accum=accum)[0] if window.size else np.nan
        if not np.isfinite(dlevel):
            failure = overscan.OverscanFailure(c, vfit[c], None, None, 'level of the horizontal overscan is not finite '
                                               '(window of {} columns)'.format(window.shape[1]))
            break
        strip -= np.float32(dlevel)
        strips.append(strip)
        aux['dlevel'].append(float(dlevel))
    # read noise per channel on the GPU (float64 accumulators); runs while the host fits
    if failure is None:
        check(lib.bbx_vos_std(ctx.h, C.byref(geom), _ptr(raw), rt, g32, _ptr(d_vfit),
                              _lib.f32x16(np.float32(aux['dlevel'])), _ptr(d_std), ctx.stream()),
              'bbx_vos_std', ctx.h)
    for c in range(16 if failure is None else failure.chan):
        try:
            data_hos = strips[c][:, :xsz]
            if tel == 'ML1':
                mask_hos = overscan.hos_mask_ml1(data_hos, data_limit)
                msr = None
            else:
                msr = mask_sat_rows[c]
                mask_hos = np.zeros(data_hos.shape, dtype=bool) | msr[None, :]
            n, mean_hos, std_hos = overscan.hos_column_stats(data_hos, mask_hos, accum=accum)
            oscan[c] = overscan.hos_fit(n, mean_hos, std_hos, msr, bg2_chan9=(tel == 'BG2' and c == 8),
                                        accum=accum)
        except Exception as e:
            failure = overscan.OverscanFailure(c, vfit[c], None, None, 'horizontal overscan: {}: {}'.format(type(e).__name__, e))
            oscan[c] = 0.0
            break
        aux['mean_hos'].append(mean_hos)
        aux['n_hos'].append(n)
    return failure, d_vfit


def os_solve(ctx, raw, header, tel, geom, data_limit=2000, accum='f32seq'):
    """os_corr (blackbox.py:6407-6879) up to the point where the overscan vectors
    are known: strip statistics on the GPU, fits on the host.  Updates [header]
    (BIAS{c}A{k}, VFITOK{c}, BIASM{c}, RDN{c}, BIASMEAN, RDNOISE, N-INFNAN)."""
    dev = ctx.device
    gain = get_par(settings.gain, tel)
    g32 = _lib.f32x16(gain)
    rt = raw_type_of(raw)
    ny_raw, nx_raw = raw.shape
    ysz, xsz = geom.ysize_chan, geom.xsize_chan
    dy, dx = ny_raw // 2, nx_raw // 8
    os_y = dy - ysz
    hos_rows = os_y - 10
    d_mean = torch.empty(16 * dy, dtype=torch.float64, device=dev)
    d_hos = torch.empty((16, hos_rows, dx), dtype=torch.float32, device=dev)
    d_ninf = torch.zeros(1, dtype=torch.int64, device=dev)
    check(lib.bbx_overscan_stats(ctx.h, C.byref(geom), _ptr(raw), rt, g32, _ptr(d_mean), _ptr(d_hos),
                                 _ptr(d_ninf), ctx.stream()), 'bbx_overscan_stats', ctx.h)
    # one device->host hop: 16*dy float64 + the hos strips (< 2 MB)
    mean_vos_col = d_mean.cpu().numpy().reshape(16, dy)
    hos = d_hos.cpu().numpy()
    header['N-INFNAN'] = (int(d_ninf.item()), 'number of pixels with infinite/nan values')
    vfit = np.empty((16, dy))
    mean_vos = np.zeros(16)
    oscan = np.zeros((16, xsz))
    aux = dict(dlevel=[], mean_hos=[], n_hos=[])
    d_std = torch.empty(16, dtype=torch.float64, device=dev)
    failure = None
    if tel == 'ML1' and overscan.fast_path_available(accum, hos.dtype):
        # MeerLICHT: a channel's fits need nothing from the device in between -- the 16 channels are solved side by
        # side (the C driver and LAPACK run without the interpreter lock); same arithmetic as the loop below
        res = [None] * 16

        def solve(c):
            try:
                res[c] = overscan.channel_solve((c, mean_vos_col[c], hos[c], ysz, xsz, settings.voscan_poldeg, tel, data_limit, accum))
            except overscan.OverscanFailure as e:
                res[c] = e
        list(_fit_threads().map(solve, range(16)))
        for c in range(16):
            r = res[c]
            if isinstance(r, overscan.OverscanFailure):
                vfit[c] = r.fit
                if r.coeffs is not None:
                    _vos_header(header, c, r.coeffs, r.ok)
                failure = r
                break
            _vos_header(header, c, r['coeffs'], r['ok'])
            vfit[c], mean_vos[c], oscan[c] = r['fit'], r['level'], r['oscan']
            aux['dlevel'].append(float(r['dlevel']))
        d_vfit = torch.from_numpy(vfit.reshape(-1)).to(dev)
        if failure is None:
            check(lib.bbx_vos_std(ctx.h, C.byref(geom), _ptr(raw), rt, g32, _ptr(d_vfit),
                                  _lib.f32x16(np.float32(aux['dlevel'])), _ptr(d_std), ctx.stream()), 'bbx_vos_std', ctx.h)
    else:
        failure, d_vfit = _os_solve_stepwise(ctx, raw, header, tel, geom, data_limit, accum, mean_vos_col, hos, vfit, mean_vos, oscan, aux,
                                             d_std, g32, rt)
    if failure is not None:
        # how far the reference's in-place os_corr had got when it raised (overscan.OverscanFailure):
        # channels before the failing one fully corrected, that one by its vertical fit, the rest untouched
        k = failure.chan
        vfit[k + 1:] = 0.0
        oscan[k:] = 0.0
        for c in range(k + 1, 16):
            for key in ['VFITOK{}'.format(c + 1)] + ['BIAS{}A{}'.format(c + 1, j) for j in range(settings.voscan_poldeg + 1)]:
                header.pop(key, None)
        failure.partial = (vfit, oscan)
        raise failure
    std_vos = d_std.cpu().numpy()
    for c in range(16):
        header['BIASM{}'.format(c + 1)] = (float(mean_vos[c]), '[e-] channel {} mean vertical overscan'.format(c + 1))
    for c in range(16):
        header['RDN{}'.format(c + 1)] = (float(std_vos[c]), '[e-] channel {} sigma (STD) vertical overscan'.format(c + 1))
    header['BIASMEAN'] = (float(np.nanmean(mean_vos)), '[e-] average all channel means vert. overscan')
    header['RDNOISE'] = (float(np.nanmean(std_vos)), '[e-] average all channel sigmas vert. overscan')
    sol = OverscanSolution()
    sol.vfit, sol.oscan = vfit, oscan
    sol.d_vfit = d_vfit
    sol.d_oscan = torch.from_numpy(oscan.reshape(-1)).to(dev)
    sol.aux = aux
    return sol


def satlevels(header, tel):
    """mask_init 4448-4463: channel saturation thresholds in e-"""
    gain = np.array(get_par(settings.gain, tel))
    bias = np.array([header['BIASM{}'.format(c + 1)][0] if isinstance(header['BIASM{}'.format(c + 1)], tuple)
                     else header['BIASM{}'.format(c + 1)] for c in range(16)])
    return np.array(get_par(settings.satlevel, tel)) * gain - bias


_SATLEV_KEYS = [('SATLEV{}'.format(c + 1), '[e-] channel {} saturation threshold'.format(c + 1)) for c in range(16)]


def calibrate(ctx, raw, sol, header, header_mask, tel, geom, mbias=None, mflat=None, bpm=None, out=None,
              satlevel_override=None):
    """gain + overscan + crop (+ master bias) + first half of mask_init (+ master
    flat) in one pass -> (data float32, mask uint8) device tensors; out: optional
    (data, mask) pair to write into instead of allocating"""
    dev = ctx.device
    ny, nx = 2 * geom.ysize_chan, 8 * geom.xsize_chan
    if tuple(raw.shape) != (geom.ny_raw, geom.nx_raw) or not raw.is_contiguous() or not raw.is_cuda:
        raise ValueError('raw frame: contiguous device tensor of shape {} expected'.format((geom.ny_raw, geom.nx_raw)))
    gain = get_par(settings.gain, tel)
    if satlevel_override is not None:
        sat = np.full(16, satlevel_override, dtype=np.float64)         # calibration frames: no saturation marking
    else:
        sat = satlevels(header, tel)
        header_mask['SATURATE'] = header['SATURATE'] = (float(np.mean(sat)), '[e-] mean saturation threshold')
        for c, (key, comment) in enumerate(_SATLEV_KEYS):
            header[key] = header_mask[key] = (round(float(sat[c]), 1), comment)
    if out is not None:
        data, mask = out
        _expect(data, torch.float32, (ny, nx), 'out data')
        _expect(mask, torch.uint8, (ny, nx), 'out mask')
    else:
        data = torch.empty((ny, nx), dtype=torch.float32, device=dev)
        mask = torch.empty((ny, nx), dtype=torch.uint8, device=dev)
    for t, name, dt in ((mbias, 'master bias', torch.float32), (mflat, 'master flat', torch.float32),
                        (bpm, 'bad pixel mask', torch.uint8)):
        if t is not None and (t.dtype != dt or tuple(t.shape) != (ny, nx) or not t.is_contiguous()):
            raise ValueError('{}: expected contiguous {} of shape {}'.format(name, dt, (ny, nx)))
    check(lib.bbx_calibrate(ctx.h, C.byref(geom), _ptr(raw), raw_type_of(raw), _lib.f32x16(gain),
                            _ptr(sol.d_vfit), _ptr(sol.d_oscan), _ptr(mbias), _ptr(mflat), _ptr(bpm),
                            _lib.f32x16(np.float32(sat)), _ptr(data), _ptr(mask), ctx.stream()),
          'bbx_calibrate', ctx.h)
    return data, mask


def mask_init_finish(ctx, mask, header, header_mask, geom, d_n=None):
    """second half of mask_init (blackbox.py:4504-4566) + fill_sat_holes; d_n: optional int32[1]
    device tensor for the NOBJ-SAT count (set by the library)"""
    _expect(mask, torch.uint8, (2 * geom.ysize_chan, 8 * geom.xsize_chan), 'mask')
    if d_n is None:
        d_n = torch.empty(1, dtype=torch.int32, device=ctx.device)
    check(lib.bbx_mask_finish(ctx.h, C.byref(geom), _ptr(mask), _ptr(d_n), ctx.stream()), 'bbx_mask_finish', ctx.h)
    return d_n


def cosmics_corr(ctx, data, header, data_mask, header_mask, tel, d_rdn16=None, d_stats=None):
    """blackbox.py:4259-4370.  In place; returns the device stats tensor
    [per-iteration counts x6, n objects, n pixels].  readnoise = header RDNOISE, or -- when
    the 16 channel sigmas are still on the device -- their nanmean taken there."""
    ny, nx = data.shape
    _expect(data, torch.float32, (ny, nx), 'data')
    _expect(data_mask, torch.uint8, (ny, nx), 'data_mask')
    readnoise = 0.0
    if d_rdn16 is None:
        hv = header['RDNOISE']
        readnoise = hv[0] if isinstance(hv, tuple) else hv
    if d_stats is None:
        d_stats = torch.empty(16, dtype=torch.int32, device=ctx.device)       # zeroed by the library
    check(lib.bbx_lacosmic(ctx.h, ny, nx, _ptr(data), _ptr(data_mask),
                           float(get_par(settings.sigclip, tel)), float(get_par(settings.sigfrac, tel)),
                           float(get_par(settings.objlim, tel)), int(get_par(settings.niter, tel)),
                           float(np.float32(readnoise)), _ptr(d_rdn16), _ptr(d_stats), ctx.stream()),
          'bbx_lacosmic', ctx.h)
    return d_stats


def detect_cosmics(ctx, data, mask, sigclip, sigfrac, objlim, niter, readnoise):
    """astroscrappy-style call on device tensors (in place) -> stats tensor"""
    ny, nx = data.shape
    _expect(data, torch.float32, (ny, nx), 'data')
    _expect(mask, torch.uint8, (ny, nx), 'mask')
    d_stats = torch.zeros(16, dtype=torch.int32, device=ctx.device)
    check(lib.bbx_lacosmic(ctx.h, ny, nx, _ptr(data), _ptr(mask), float(sigclip), float(sigfrac), float(objlim),
                           int(niter), float(np.float32(readnoise)), _ptr(None), _ptr(d_stats), ctx.stream()),
          'bbx_lacosmic', ctx.h)
    return d_stats


def read_crosstalk(path):
    """ASCII table 'victim source correction', 1-based channels -> coeffs[source, victim]
    (blackbox.py:7157-7198)"""
    coeffs = np.zeros((16, 16))
    with open(path) as f:
        first = f.readline().split()
        if first[:3] != ['victim', 'source', 'correction']:
            v, s, c = first[:3]
            coeffs[int(s) - 1, int(v) - 1] = float(c)
        for line in f:
            p = line.split()
            if len(p) >= 3:
                coeffs[int(p[1]) - 1, int(p[0]) - 1] = float(p[2])
    return coeffs


def xtalk_corr(ctx, data, coeffs, data_mask, geom):
    """blackbox.py:7138-7258, in place on the device"""
    shape = (2 * geom.ysize_chan, 8 * geom.xsize_chan)
    _expect(data, torch.float32, shape, 'data')
    _expect(data_mask, torch.uint8, shape, 'data_mask')
    cf = (C.c_double * 256)(*[float(v) for v in np.asarray(coeffs, dtype=np.float64).reshape(-1)])
    check(lib.bbx_xtalk(ctx.h, C.byref(geom), _ptr(data), _ptr(data_mask), cf, ctx.stream()), 'bbx_xtalk', ctx.h)


NL_MAXKNOTS = 256


def spline_tck(spl):
    """(t, c, k) of a scipy UnivariateSpline (what its __call__ hands to FITPACK splev) or of
    a (t, c, k) tuple"""
    if isinstance(spl, (tuple, list)):
        t, c, k = spl
    else:
        t, c, k = spl._eval_args
    return np.asarray(t, np.float64), np.asarray(c, np.float64), int(k)


def read_nonlin_splines(path):
    """the reference's nonlin_corr_file: a pickled list of 16 scipy spline objects
    (blackbox.py:7400-7401)"""
    import pickle
    with open(path, 'rb') as f:
        return pickle.load(f)


def set_nonlin(ctx, splines):
    """load the 16 per-channel non-linearity splines into the context (None switches the
    correction off); while set, calibrate() applies nonlin_corr between the overscan and the
    master-bias steps like blackbox_reduce (blackbox.py:1604-1624)"""
    if splines is None:
        check(lib.bbx_nonlin_set(ctx.h, 0, None, None, None), 'bbx_nonlin_set', ctx.h)
        return
    if len(splines) != 16:
        raise ValueError('need one spline per channel (16)')
    tck = [spline_tck(s) for s in splines]
    k = tck[0][2]
    if any(x[2] != k for x in tck):
        raise ValueError('splines of different degree')
    n = (C.c_int32 * 16)()
    t = np.zeros((16, NL_MAXKNOTS)); c = np.zeros((16, NL_MAXKNOTS))
    for i, (ti, ci, _) in enumerate(tck):
        if ti.size > NL_MAXKNOTS:
            raise ValueError('spline with more than %d knots' % NL_MAXKNOTS)
        n[i] = ti.size
        t[i, :ti.size] = ti
        c[i, :ci.size] = ci[:ti.size]
    check(lib.bbx_nonlin_set(ctx.h, k, n, t.ctypes.data_as(_lib._pd), c.ctypes.data_as(_lib._pd)), 'bbx_nonlin_set', ctx.h)


def nonlin_corr(ctx, data, geom, tel, splines=None):
    """blackbox.py:7394-7437 in place on an overscan-corrected device frame (the splines
    are those of the last set_nonlin unless given)"""
    if splines is not None:
        set_nonlin(ctx, splines)
    g32 = _lib.f32x16(get_par(settings.gain, tel))
    check(lib.bbx_nonlin_corr(ctx.h, C.byref(geom), _ptr(data), g32, ctx.stream()), 'bbx_nonlin_corr', ctx.h)
    return data


SAT_THETA_DEG = np.arange(2, 178, 0.5, dtype=float)      # acstools.satdet: np.radians(np.arange(2, 178, 0.5))
NTHETA_SAT = SAT_THETA_DEG.size
SAT_SIGMA = 3.0                                           # sat_detect: detsat(..., sigma=3, ...)

_SAT_CS = None


def sat_cos_sin():
    """cos / sin table of the Hough angles (float64), shared by the device code and the oracle"""
    global _SAT_CS
    if _SAT_CS is None:
        th = np.radians(SAT_THETA_DEG)
        cs = np.empty(2 * NTHETA_SAT)
        cs[0::2], cs[1::2] = np.cos(th), np.sin(th)
        _SAT_CS = (cs, cs.ctypes.data_as(C.POINTER(C.c_double)))
    return _SAT_CS[1]


_SAT_GW = None


def sat_gauss_weights():
    """scipy.ndimage's Gaussian kernel for Canny's sigma, centre first: radius int(4 sigma + 0.5),
    exp(-0.5 / sigma^2 x^2) normalised by its sum over -radius .. radius (float64) -> (pointer, radius)"""
    global _SAT_GW
    if _SAT_GW is None:
        r = int(4.0 * SAT_SIGMA + 0.5)
        x = np.arange(-r, r + 1)
        w = np.exp(-0.5 / (SAT_SIGMA * SAT_SIGMA) * x ** 2)
        w = np.ascontiguousarray((w / w.sum())[r:])
        _SAT_GW = (w, w.ctypes.data_as(C.POINTER(C.c_double)), r)
    return _SAT_GW[1], _SAT_GW[2]


def sat_detect(ctx, data, header, data_mask, header_mask):
    """blackbox.py:4163-4254 (classical path; deterministic detector, see oracle/sattrail.py):
    adds bit 16 to data_mask in place, sets NSATS.  Returns (nsats tensor, info tensor)."""
    ny, nx = data.shape
    _expect(data, torch.float32, (ny, nx), 'data')
    _expect(data_mask, torch.uint8, (ny, nx), 'data_mask')
    d_n = torch.zeros(1, dtype=torch.int32, device=ctx.device)
    d_info = torch.zeros(8, dtype=torch.float32, device=ctx.device)
    gw, gr = sat_gauss_weights()
    check(lib.bbx_sat_trails(ctx.h, ny, nx, _ptr(data), _ptr(data_mask), sat_cos_sin(),
                             NTHETA_SAT, gw, gr, _ptr(d_n), _ptr(d_info), ctx.stream()), 'bbx_sat_trails', ctx.h)
    return d_n, d_info


def mask_header(ctx, data_mask, header_mask):
    """blackbox.py:4601-4620"""
    _expect(data_mask, torch.uint8, data_mask.shape, 'data_mask')
    d_c = torch.zeros(6, dtype=torch.int64, device=ctx.device)
    check(lib.bbx_mask_counts(ctx.h, data_mask.numel(), _ptr(data_mask), _ptr(d_c), ctx.stream()),
          'bbx_mask_counts', ctx.h)
    fill_mask_header(header_mask, d_c.cpu().numpy())


def fill_mask_header(header_mask, counts):
    """the M-* keywords of mask_header from the six pixel counts (bad, edge, saturated,
    saturated-connected, satellite trail, cosmic ray)"""
    text = (('bad', 'BP'), ('edge', 'EP'), ('saturated', 'SP'), ('saturated-connected', 'SCP'),
            ('satellite trail', 'STP'), ('cosmic ray', 'CRP'))
    for (mask_type, t), n in zip(text, counts):
        value = settings.mask_value[mask_type]
        header_mask['M-{}'.format(t)] = (True, '{} pixels included in mask?'.format(mask_type))
        header_mask['M-{}VAL'.format(t)] = (value, 'value added to mask for {} pixels'.format(mask_type))
        header_mask['M-{}NUM'.format(t)] = (int(n), 'number of {} pixels'.format(mask_type))


def edge_fill(ctx, data, data_mask, geom):
    """blackbox.py:1959-1974"""
    shape = (2 * geom.ysize_chan, 8 * geom.xsize_chan)
    _expect(data, torch.float32, shape, 'data')
    _expect(data_mask, torch.uint8, shape, 'data_mask')
    d_med = torch.empty(16, dtype=torch.float32, device=ctx.device)
    check(lib.bbx_edge_fill(ctx.h, C.byref(geom), _ptr(data), _ptr(data_mask), _ptr(d_med), ctx.stream()),
          'bbx_edge_fill', ctx.h)
    return d_med


def count_objects(ctx, data_mask, bit):
    ny, nx = data_mask.shape
    d_n = torch.zeros(1, dtype=torch.int32, device=ctx.device)
    check(lib.bbx_count_objects(ctx.h, ny, nx, _ptr(data_mask), int(bit), _ptr(d_n), ctx.stream()),
          'bbx_count_objects', ctx.h)
    return d_n


def image_to_device(ctx, path, dtype):
    """a FITS image (master frame, reference image, mask) -> device tensor of [dtype].  float32 and uint8 files go up as
    their bytes, float32 words are put into host order on the device (bbx_be32); anything else is converted on the host"""
    from . import fitsio
    got = fitsio.read_image_file_order(path) if str(path).endswith(('.fits', '.fit')) else None
    if got is not None:
        data = got[0]
        if data.dtype == np.dtype('>f4') and np.dtype(dtype) == np.float32:
            t = torch.from_numpy(data.view(np.int32)).to(ctx.device)       # (the bytes; int32 is only the carrier)
            check(lib.bbx_be32(_ptr(t), _ptr(t), t.numel(), ctx.stream()), 'bbx_be32')
            return t.view(torch.float32)
        if data.dtype == np.uint8 and np.dtype(dtype) == np.uint8:
            return torch.from_numpy(data).to(ctx.device)
    return torch.from_numpy(np.ascontiguousarray(fitsio.read_image(path, dtype=dtype))).to(ctx.device)


def hval(header, key):
    v = header[key]
    return v[0] if isinstance(v, tuple) else v


# ---- error attribution per step --------------------------------------------------------------
STEP_SLOTS = ('calibrate', 'mask', 'cosmics', 'xtalk', 'sat', 'finish', 'bkg', 'zogy')


def step_mark(ctx, d_steps, step):
    """enqueue: d_steps[slot of step] <- device error flags raised since the previous mark (bbx_step_mark)"""
    k = STEP_SLOTS.index(step)
    check(lib.bbx_step_mark(ctx.h, C.c_void_p(d_steps.data_ptr() + 4 * k), ctx.stream()), 'bbx_step_mark', ctx.h)


def zero_overscan_solution(ctx, header, geom, partial=None, header_only=False):
    """the reference's fallback when os_corr raises (blackbox.py:1546-1585): adopt an overscan of
    zero for all channels -- the data sections are only cropped -- with BIASM{c} = 0, RDN{c} = 10,
    BIASMEAN = 0, RDNOISE = 10.  partial = (vfit, oscan): what the reference's in-place os_corr had
    subtracted from the channels it got through before it raised (overscan.OverscanFailure) -- the
    crop is taken from that half-processed array, so those channels keep their correction"""
    dy = geom.ny_raw // 2
    for c in range(16):
        header['BIASM{}'.format(c + 1)] = (0.0, '[e-] channel {} mean vertical overscan'.format(c + 1))
    for c in range(16):
        header['RDN{}'.format(c + 1)] = (10.0, '[e-] channel {} sigma (STD) vertical overscan'.format(c + 1))
    header['BIASMEAN'] = (0.0, '[e-] average all channel means vert. overscan')
    header['RDNOISE'] = (10.0, '[e-] average all channel sigmas vert. overscan')
    if header_only:
        return None
    sol = OverscanSolution()
    if partial is not None:
        sol.vfit, sol.oscan = (np.ascontiguousarray(a, np.float64) for a in partial)
        sol.d_vfit = torch.from_numpy(sol.vfit.reshape(-1)).to(ctx.device)
        sol.d_oscan = torch.from_numpy(sol.oscan.reshape(-1)).to(ctx.device)
    else:
        sol.vfit, sol.oscan = np.zeros((16, dy)), np.zeros((16, geom.xsize_chan))
        sol.d_vfit = torch.zeros(16 * dy, dtype=torch.float64, device=ctx.device)
        sol.d_oscan = torch.zeros(16 * geom.xsize_chan, dtype=torch.float64, device=ctx.device)
    sol.aux = None
    return sol


def apply_step_errors(header, header_mask, errs, log=None):
    """device-side failures of the asynchronous stages (list overflow, non-convergence), read
    per step after the frame's synchronisation -> the reference's convention (blackbox.py:
    1750-1761, 1866-1878, 1897-1912, 1919-1952): flag the step, keep going with what there is"""
    def bad(step):
        return int(errs[STEP_SLOTS.index(step)]) != 0
    if bad('calibrate') or bad('mask'):
        header['MASK-P'] = (False, 'mask image created?')
    if bad('cosmics'):
        header['COSMIC-P'] = (False, 'corrected for cosmic rays?')
        header['NCOSMICS'] = header_mask['NCOSMICS'] = ('None', '[/s] number of cosmic rays identified')
    if bad('xtalk'):
        header['XTALK-P'] = (False, 'corrected for crosstalk?')
    if bad('sat'):
        header['SAT-P'] = (False, 'processed for satellite trails?')
        header['NSATS'] = header_mask['NSATS'] = ('None', 'number of satellite trails identified')
    failed = [s for s in STEP_SLOTS if bad(s)]
    if failed and log is not None:
        log.error('device-side error flags %s in step(s) %s', [int(e) for e in errs], failed)
    return failed


def reduce_object(ctx, raw, header, tel, mflat=None, mbias=None, bpm=None, xtalk_coeffs=None,
                  exptime=None, ysize_chan=None, xsize_chan=None, do_cosmics=True, crmask_override=None,
                  accum='f32seq', stages=None, detect_sats=True, nonlin_splines=None, imgtype='object', log=None):
    """The hot path of blackbox_reduce (blackbox.py:1451-1974): raw device tensor ->
    (data, mask, header, header_mask).  imgtype 'object' runs every step; 'flat' stops after the
    master-bias step with the bad-pixel mask as its mask (mask_init with imgtype 'flat' copies the
    BPM only, 4386-4405; no saturation marking, flat division, cosmics, crosstalk, trails, edge
    fill); 'bias' / 'dark' stop after the overscan step (1627-1637).
    Failures of a step -- an exception of its host part or a device-side error flag -- follow the
    reference convention: `<STEP>-P = False`, log, carry on with what there is."""
    header_mask = {}
    geom = geometry(raw.shape, ysize_chan, xsize_chan)
    d_steps = torch.zeros(len(STEP_SLOTS), dtype=torch.int32, device=ctx.device)

    def logexc(what):
        if log is not None:
            log.exception('exception was raised during [%s]', what)

    try:
        gain_ok = False
        gain_corr(header, tel)
        gain_ok = True
    except Exception:
        logexc('gain_corr')
    header['GAIN'] = (1.0, '[e-/ADU] effective gain all channels')
    header['GAIN-P'] = (gain_ok, 'corrected for gain?')
    try:
        sol = os_solve(ctx, raw, header, tel, geom, accum=accum)
        os_ok = True
    except Exception as e:
        # os_corr failed: adopt an overscan of zero for all channels (blackbox.py:1537-1585)
        logexc('os_corr; adopting an overscan of zero for all channels')
        sol = zero_overscan_solution(ctx, header, geom, partial=getattr(e, 'partial', None))
        os_ok = False
    header['OS-P'] = (os_ok, 'corrected for overscan?')
    # non-linearity correction (off upstream: set_bb.correct_nonlin False): done inside the fused
    # calibration pass when splines are given
    header['NONLIN-P'] = (False, 'corrected for non-linearity?')
    if imgtype != 'bias' and nonlin_splines is not None:
        try:
            set_nonlin(ctx, nonlin_splines)
            header['NONLIN-P'] = (True, 'corrected for non-linearity?')
        except Exception:
            logexc('nonlin_corr')
            set_nonlin(ctx, None)
    else:
        set_nonlin(ctx, None)
    is_object = imgtype == 'object'
    use_bias = imgtype in ('object', 'flat') and mbias is not None and bool(get_par(settings.subtract_mbias, tel))
    use_flat = is_object and mflat is not None
    if is_object:
        data, mask = calibrate(ctx, raw, sol, header, header_mask, tel, geom,
                               mbias=mbias if use_bias else None, mflat=mflat if use_flat else None, bpm=bpm)
    else:
        # calibration frames: no saturation marking (mask_init marks saturated pixels for object
        # frames only), mask = the bad-pixel mask as it is
        data, _ = calibrate(ctx, raw, sol, {k: v for k, v in header.items()}, {}, tel, geom,
                            mbias=mbias if use_bias else None, mflat=None, bpm=None, satlevel_override=np.inf)
        mask = bpm.clone() if bpm is not None else torch.zeros(data.shape, dtype=torch.uint8, device=ctx.device)
    step_mark(ctx, d_steps, 'calibrate')
    header['MBIAS-P'] = (bool(use_bias), 'corrected for master bias?')
    if not is_object:
        ctx.sync()
        if imgtype == 'flat':
            header['MASK-P'] = (True, 'mask image created?')
        return data, mask, header, header_mask
    header['MFLAT-P'] = (bool(use_flat), 'corrected for master flat?')
    d_nobj = None
    try:
        d_nobj = mask_init_finish(ctx, mask, header, header_mask, geom)
        header['MASK-P'] = (True, 'mask image created?')
    except _lib.BBXError:
        logexc('mask_init')
        header['MASK-P'] = (False, 'mask image created?')
    step_mark(ctx, d_steps, 'mask')
    d_stats = None
    if crmask_override is not None:
        # test hook: take the cosmic-ray pixels as given instead of detecting them
        mask |= (crmask_override.to(mask.device).to(torch.uint8) * 2)
    elif do_cosmics:
        try:
            d_stats = cosmics_corr(ctx, data, header, mask, header_mask, tel)
            header['COSMIC-P'] = (True, 'corrected for cosmic rays?')
        except _lib.BBXError:
            logexc('cosmics_corr')
            d_stats = None
            header['NCOSMICS'] = ('None', '[/s] number of cosmic rays identified')
            header['COSMIC-P'] = (False, 'corrected for cosmic rays?')
    step_mark(ctx, d_steps, 'cosmics')
    if xtalk_coeffs is not None:
        try:
            xtalk_corr(ctx, data, xtalk_coeffs, mask, geom)
            header['XTALK-P'] = (True, 'corrected for crosstalk?')
        except (_lib.BBXError, ValueError):
            logexc('xtalk_corr')
            header['XTALK-P'] = (False, 'corrected for crosstalk?')
    step_mark(ctx, d_steps, 'xtalk')
    if stages is not None:
        stages['data_xtalk'] = data.clone()
    d_nsats = None
    if detect_sats and get_par(settings.detect_sats, tel):
        try:
            d_nsats, _ = sat_detect(ctx, data, header, mask, header_mask)
            header['SAT-P'] = (True, 'processed for satellite trails?')
        except _lib.BBXError:
            logexc('sat_detect')
            d_nsats = None
            header['NSATS'] = ('None', 'number of satellite trails identified')
            header['SAT-P'] = (False, 'processed for satellite trails?')
    step_mark(ctx, d_steps, 'sat')
    mask_header(ctx, mask, header_mask)
    edge_fill(ctx, data, mask, geom)
    step_mark(ctx, d_steps, 'finish')
    ctx.sync()
    if d_nobj is not None:
        nobj = int(d_nobj.item())
        header_mask['NOBJ-SAT'] = header['NOBJ-SAT'] = (nobj, 'number of saturated objects')
    if d_nsats is not None:
        header['NSATS'] = header_mask['NSATS'] = (int(d_nsats.item()), 'number of satellite trails identified')
    if d_stats is not None:
        st = d_stats.cpu().numpy()
        t = float(exptime) if exptime else 1.0
        header['NCOSMICS'] = header_mask['NCOSMICS'] = (float(st[6]) / t, '[/s] number of cosmic rays identified')
        header['NCRPIX'] = (int(st[7]), 'number of cosmic-ray pixels')
    apply_step_errors(header, header_mask, d_steps.cpu().numpy(), log)
    return data, mask, header, header_mask
