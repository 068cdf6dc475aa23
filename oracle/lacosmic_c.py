"""TEST INFRASTRUCTURE -- ctypes binding of oracle/lacosmic_c.c (the C twin of oracle/lacosmic.py; same results bit for bit,
tests/test_lacosmic_oracle.py).  Used by bench.py's cpu_baseline leg: astroscrappy, which the reference runs here
(blackbox.py:4323-4332), is compiled C with OpenMP as well."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'liblacosmic_c.so')
_SRC = os.path.join(_HERE, 'lacosmic_c.c')


def build():
    """gcc the C twin when the library is missing or older than its source"""
    if not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(['gcc', '-O3', '-fopenmp', '-ffp-contract=off', '-fPIC', '-shared', '-o', _SO, _SRC, '-lm'])
    return _SO


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.lac_detect_cosmics.restype = C.c_int
        _lib.lac_detect_cosmics.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return _lib


def detect_cosmics(indat, inmask, sigclip, sigfrac, objlim, niter, readnoise, return_iters=False, nthreads=1):
    """same call as lacosmic.detect_cosmics -> (crmask bool, cleanarr float32[, pixels flagged per iteration])"""
    lib = _load()
    a = np.ascontiguousarray(indat, np.float32)
    m = np.ascontiguousarray(np.asarray(inmask) != 0, np.uint8)
    ny, nx = a.shape
    cr = np.empty((ny, nx), np.uint8)
    clean = np.empty((ny, nx), np.float32)
    nit = np.full(max(1, niter), -1, np.int64)
    rc = lib.lac_detect_cosmics(a.ctypes.data, m.ctypes.data, ny, nx, np.float32(sigclip), np.float32(sigfrac), np.float32(objlim), int(niter),
                                np.float32(readnoise), cr.ctypes.data, clean.ctypes.data, nit.ctypes.data, int(nthreads))
    if rc:
        raise MemoryError('lac_detect_cosmics')
    if return_iters:
        return cr.astype(bool), clean, [int(v) for v in nit if v >= 0]
    return cr.astype(bool), clean
