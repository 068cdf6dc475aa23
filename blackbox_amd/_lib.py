"""ctypes binding of libbbx_hip.so (the C ABI declared in include/bbx.h).

The library is mandatory: if it is missing or a symbol cannot be resolved the
import of this module raises -- there is no CPU fallback in the product path.
"""
import ctypes as C
import os
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
# BBX_LIB_PATH: a scratch build of the library (kernel-variant experiments, tools/exp/zvar.sh); the product is the in-tree one
LIB_PATH = os.environ.get('BBX_LIB_PATH') or os.path.join(_HERE, 'libbbx_hip.so')

BBX_RAW_U16, BBX_RAW_F32 = 0, 1


class BBXError(RuntimeError):
    def __init__(self, code, what, detail=''):
        self.code = code
        msg = '{} failed: {} ({})'.format(what, _strerror(code), code)
        if detail:
            msg += ' -- ' + detail
        RuntimeError.__init__(self, msg)


class Geom(C.Structure):
    _fields_ = [('ny_raw', C.c_int32), ('nx_raw', C.c_int32),
                ('ysize_chan', C.c_int32), ('xsize_chan', C.c_int32)]


class SplineImage(C.Structure):
    """bbx_spline_image (include/bbx.h): a mini image as the B-spline coefficients of its edge-padded patches"""
    _fields_ = [('d_coef', C.c_void_p), ('nby', C.c_int32), ('nbx', C.c_int32), ('cy', C.c_int32), ('cx', C.c_int32),
                ('box', C.c_int32), ('npad', C.c_int32)]


if not os.path.isfile(LIB_PATH):
    raise ImportError('{} not found: build it with `make` (hipcc, gfx950); the '
                      'reduction has no CPU fallback'.format(LIB_PATH))
# PyTorch (device memory, streams) ships its own copy of the HIP runtime.  It has to be the first one in the process: loaded
# after this library -- which names libamdhip64 of /opt/rocm -- it ends up sharing a runtime of another release and no
# device is found ("no ROCm-capable device is detected" from bbx_ctx_create).  So: torch first, whenever it is there.
try:
    import torch  # noqa: F401
except ImportError:      # the C ABI can be used without it (include/bbx.h); the host layer of this package cannot
    pass
lib = C.CDLL(LIB_PATH)

_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
_pf = C.POINTER(C.c_float)
_pd = C.POINTER(C.c_double)
_pg = C.POINTER(Geom)

# every symbol include/bbx.h declares, with its signature
SIGNATURES = {
    'bbx_ctx_create': (_i, [_i, C.POINTER(_vp)]),
    'bbx_ctx_destroy': (None, [_vp]),
    'bbx_strerror': (C.c_char_p, [_i]),
    'bbx_last_hip_error': (C.c_char_p, [_vp]),
    'bbx_version': (_i, []),
    'bbx_sync': (_i, [_vp, _vp]),
    'bbx_set_option': (_i, [_vp, _i, _i]),
    'bbx_wait': (_i, [_vp, _vp]),
    'bbx_copy_kernel': (_i, [_vp, _vp, C.c_size_t, _vp]),
    'bbx_event_wait': (_i, [_vp, _i]),
    'bbx_build_flags': (_i, []),
    'bbx_step_mark': (_i, [_vp, _vp, _vp]),
    'bbx_event_create': (_i, [C.POINTER(C.c_void_p)]),
    'bbx_event_destroy': (None, [_vp]),
    'bbx_event_record': (_i, [_vp, _vp]),
    'bbx_event_query': (_i, [_vp]),
    'bbx_stream_wait_event': (_i, [_vp, _vp]),
    'bbx_copy_async': (_i, [_vp, _vp, C.c_size_t, _i, _vp]),
    'bbx_profile_enable': (_i, [_vp, _i]),
    'bbx_profile_read': (_i, [_vp, _pd, C.POINTER(C.c_int32), _i]),
    'bbx_overscan_stats': (_i, [_vp, _pg, _vp, _i, _pf, _vp, _vp, _vp, _vp]),
    'bbx_vos_std': (_i, [_vp, _pg, _vp, _i, _pf, _vp, _pf, _vp, _vp]),
    'bbx_satcol_counts': (_i, [_vp, _pg, _vp, _i, _pf, _vp, _pf, _i, _i, _vp, _vp]),
    'bbx_calibrate': (_i, [_vp, _pg, _vp, _i, _pf, _vp, _vp, _vp, _vp, _vp, _pf, _vp, _vp, _vp]),
    'bbx_rect_stats': (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    'bbx_rect_clipped_stats': (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, C.c_double, _i, _i, _vp, _vp]),
    'bbx_frame_clipped_stats': (_i, [_vp, _i, _i, _vp, _vp, _i, C.c_double, _i, _i, _vp, _vp]),
    'bbx_fpack_tile_stride': (C.c_size_t, [_i, _i]),
    'bbx_fpack_tiles': (_i, [_vp, _i, _i, _vp, _i, _f, _i, _vp, _vp, _vp, _vp]),
    'bbx_fpack_gather': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'bbx_fpack_body': (_i, [_vp, _i, _i, _vp, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, C.c_longlong, _vp, _i, _vp]),
    'bbx_fpack_body_scaled': (_i, [_vp, _i, _i, _vp, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, C.c_longlong, _vp, _i, _f, _vp]),
    'bbx_raw_be16': (_i, [_vp, _vp, C.c_size_t, _vp]),
    'bbx_be32': (_i, [_vp, _vp, C.c_size_t, _vp]),
    'bbx_funpack_tiles': (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    'bbx_coadd_prep': (_i, [_vp, C.c_int64, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    'bbx_resample_lanczos3': (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    'bbx_coadd_combine': (_i, [_vp, _i, C.c_int64, _vp, _vp, C.c_int64, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    'bbx_clipped2mask': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _f, _i, C.POINTER(C.c_int),
                          C.POINTER(C.c_float), C.POINTER(C.c_int), _vp, _vp, _vp, _vp]),
    'bbx_psf_model': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'bbx_rect_scale': (_i, [_vp, _i, _i, _i, _vp, _f, _i, _vp]),
    'bbx_nonlin_set': (_i, [_vp, _i, C.POINTER(C.c_int32), _pd, _pd]),
    'bbx_nonlin_corr': (_i, [_vp, _pg, _vp, _pf, _vp]),
    'bbx_mask_finish': (_i, [_vp, _pg, _vp, _vp, _vp]),
    'bbx_lacosmic': (_i, [_vp, _i, _i, _vp, _vp, _f, _f, _f, _i, _f, _vp, _vp, _vp]),
    'bbx_xtalk': (_i, [_vp, _pg, _vp, _vp, _pd, _vp]),
    'bbx_mask_counts': (_i, [_vp, C.c_int64, _vp, _vp, _vp]),
    'bbx_edge_fill': (_i, [_vp, _pg, _vp, _vp, _vp, _vp]),
    'bbx_median_stack': (_i, [_vp, C.c_int64, _i, C.POINTER(_vp), _pf, _vp, _i, _vp, _vp]),
    'bbx_canny_edge_map': (_i, [_vp, _i, _i, _vp, _pd, _i, C.c_double, C.c_double, _i, _vp, _vp, _vp]),
    'bbx_sat_trails': (_i, [_vp, _i, _i, _vp, _vp, _pd, _i, _pd, _i, _vp, _vp, _vp]),
    'bbx_bkg_boxstats': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    'bbx_mini_fill_filter': (_i, [_vp, _i, _i, _vp, _vp]),
    'bbx_spline_prefilter': (_i, [_vp, _i, _i, _i, _i, _i, C.c_double, C.c_double, _vp, _vp, _vp]),
    'bbx_mini_median': (_i, [_vp, _i, _vp, _vp, _vp]),
    'bbx_zoom_candidates': (_i, [_vp, _vp, C.c_double]),
    'bbx_spline_zoom': (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'bbx_spline_zoom_sub': (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'bbx_variance': (_i, [_vp, C.c_int64, _vp, _vp, _vp, _vp]),
    'bbx_embed_psf': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    'bbx_cut_subimages': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'bbx_stitch_subimages': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'bbx_zogy_subimages': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _pf, _vp, _vp, _vp, _vp, _vp, _vp]),
    'bbx_zogy_frame_supported': (_i, [_i]),
    'bbx_zogy_candidates': (_i, [_vp, _f]),
    'bbx_zogy_frame': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _pf, _vp, _vp, _vp, _vp, _vp, _vp]),
    'bbx_zogy_frame_mini': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _pf, _vp, _vp, _vp, _vp, _vp, _vp]),
    'bbx_psf_optflux_mini': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'bbx_psf_optflux': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'bbx_psf_optflux_sigma': (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'bbx_find_peaks': (_i, [_vp, _i, _i, _vp, _f, _i, _vp, _vp, _vp, _vp]),
    'bbx_count_objects': (_i, [_vp, _i, _i, _vp, _i, _vp, _vp]),
}
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError if the export is missing
    _fn.restype = _res
    _fn.argtypes = _args


def _strerror(code):
    return lib.bbx_strerror(int(code)).decode()


WAIT_SLEEP_US = int(os.environ.get('BBX_WAIT_SLEEP_US', '50'))       # host waits of the multi-threaded paths: poll + sleep (0: spin)


WAIT_STATS = [0.0, 0]
KERNEL_COPY_MAX = int(os.environ.get('BBX_KERNEL_COPY_MAX', str(4 << 20)))    # fetch / push: copies up to this size go by a kernel


def _thread_state(ctx):
    """the calling thread's staging buffers of this context (fetch's pinned buffer, push's pinned ring): a context is driven
    from several threads -- the caller's and lane 0 of a pipeline share one -- and each moves its own small tensors"""
    import threading
    tls = ctx.__dict__.get('_tls')
    if tls is None:
        tls = ctx.__dict__.setdefault('_tls', threading.local())
    return tls.__dict__


def push(ctx, *arrays):
    """numpy arrays -> device tensors without a blocking pageable copy: the values go into the thread's pinned upload
    buffer and a kernel on the current stream reads them from there (stream-ordered).  The buffer is a ring of 64 slices; a
    slice carries the event recorded behind the kernel that read it last, and is written again only when that has
    completed (a caller pushing in a loop on a backed-up stream waits there instead of overwriting an upload in flight)"""
    import numpy as np
    import torch
    outs = []
    sp = ctx.stream()
    for a in arrays:
        a = np.ascontiguousarray(a)
        n = a.nbytes
        t = torch.empty(a.shape, dtype=torch.from_numpy(a[:0].reshape(-1)).dtype, device=ctx.device)
        if n == 0:
            outs.append(t)
            continue
        if n > KERNEL_COPY_MAX // 4:
            t.copy_(torch.from_numpy(a))
            outs.append(t)
            continue
        st = _thread_state(ctx)
        ring = st.get('push_ring')
        if ring is None:
            ring = st['push_ring'] = [torch.empty(64 * (KERNEL_COPY_MAX // 4), dtype=torch.uint8, pin_memory=True), 0, [None] * 64]
        slot = ring[1] % 64
        ring[1] += 1
        ev = ring[2][slot]
        if ev is not None and not ev.query():
            wait_event(ev)
        off = slot * (KERNEL_COPY_MAX // 4)
        ring[0].numpy()[off:off + n] = a.reshape(-1).view(np.uint8)
        check(lib.bbx_copy_kernel(C.c_void_p(t.data_ptr()), C.c_void_p(ring[0].data_ptr() + off), n, sp), 'bbx_copy_kernel')
        if ev is None:
            ev = ring[2][slot] = torch.cuda.Event()
        ev.record()                                               # (the current stream: the one the kernel went to)
        outs.append(t)
    return outs if len(outs) > 1 else outs[0]


def wait_event(ev, sleep_us=None):
    """wait for a torch.cuda.Event without spinning on a core (bbx_event_wait)"""
    us = WAIT_SLEEP_US if sleep_us is None else sleep_us
    if us <= 0:
        ev.synchronize()
        return
    h = ev.cuda_event
    if not h:
        return                                  # never recorded: nothing to wait for
    check(lib.bbx_event_wait(C.c_void_p(h), int(us)), 'bbx_event_wait')


def fetch(ctx, *tensors, check_device_errors=False):
    """device tensors -> numpy arrays (copies) with ONE host wait, which sleeps between polls when the context was told to
    (BBX_OPT_WAIT_SLEEP_US): asynchronous copies into the context's pinned staging buffer on the current stream, bbx_wait.
    What `.cpu().numpy()` does with a spinning hipStreamSynchronize per tensor."""
    import torch
    ts = [t.contiguous() for t in tensors]
    sizes = [(t.numel() * t.element_size() + 63) // 64 * 64 for t in ts]
    total = max(64, sum(sizes))
    st = _thread_state(ctx)
    pin = st.get('fetch_pin')
    if pin is None or pin.numel() < total:
        pin = st['fetch_pin'] = torch.empty(int(total * 1.5) + 4096, dtype=torch.uint8, pin_memory=True)
    views, off = [], 0
    sp = ctx.stream()
    for t, nb in zip(ts, sizes):
        n = t.numel() * t.element_size()
        v = pin[off:off + n].view(t.dtype).view(t.shape)
        if n and n <= KERNEL_COPY_MAX:
            # by a kernel, not by the copy engine: these few bytes must not queue behind the output stage's 100 MB transfers
            check(lib.bbx_copy_kernel(C.c_void_p(pin.data_ptr() + off), C.c_void_p(t.data_ptr()), n, sp), 'bbx_copy_kernel')
        elif n:
            v.copy_(t, non_blocking=True)
        views.append(v)
        off += nb
    t0 = time.perf_counter()
    if check_device_errors:
        # the same wait, and the context's device error word read with it (bbx_sync): a list that overflowed, a PSF window
        # that does not hold raise here, as they would at a ctx.sync() -- without a host wait of their own
        check(lib.bbx_sync(ctx.h, sp), 'bbx_sync', ctx.h)
    else:
        check(lib.bbx_wait(ctx.h, sp), 'bbx_wait', ctx.h)
    WAIT_STATS[0] += time.perf_counter() - t0; WAIT_STATS[1] += 1          # (diagnostics: seconds / calls of host waits in fetch)
    out = [v.numpy().copy() for v in views]
    return out if len(out) > 1 else out[0]


def f32x16(values):
    arr = (C.c_float * 16)(*[float(v) for v in values])
    return arr


def check(code, what, ctx=None):
    if code != 0:
        detail = ''
        if ctx is not None and code == -2:
            detail = lib.bbx_last_hip_error(ctx).decode()
        raise BBXError(code, what, detail)
