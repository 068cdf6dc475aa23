"""CPU: the C-ABI library loads and exports every symbol include/bbx.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'bbx.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(bbx_[a-z0-9_]+)\s*\(', text)))


def test_header_and_library_agree():
    syms = declared_symbols()
    assert len(syms) >= 15
    lib = ctypes.CDLL(os.path.join(ROOT, 'blackbox_amd', 'libbbx_hip.so'))
    for s in syms:
        assert hasattr(lib, s), 'missing export ' + s
    from blackbox_amd import _lib
    assert sorted(_lib.SIGNATURES) == syms          # the ctypes table covers the whole header
    assert _lib.lib.bbx_version() >= 100
    assert _lib.lib.bbx_strerror(-4).decode().startswith('device work list')


def test_bad_arguments_are_rejected_without_gpu():
    from blackbox_amd import _lib
    # NULL ctx / NULL pointers: argument errors, never a crash
    assert _lib.lib.bbx_sync(None, None) == -1
    g = _lib.Geom(10600, 12000, 5280, 1320)
    assert _lib.lib.bbx_mask_finish(None, ctypes.byref(g), None, None, None) == -1
    assert _lib.lib.bbx_lacosmic(None, 100, 100, None, None, 15.0, 0.01, 3.0, 3, 8.0, None, None, None) == -1
    # the rows added later: co-add, clip-log masks, options, stream plumbing
    L = _lib.lib
    assert L.bbx_coadd_prep(None, 100, None, None, None, None, 0, 32, None, None) == -1
    assert L.bbx_resample_lanczos3(None, 64, 64, None, None, 64, 64, None, 3, 3, 32, 1.0, None, None, None) == -1
    assert L.bbx_coadd_combine(None, 3, 100, None, None, 100, 0, 4.0, 0.3, None, None, None, None, None, None) == -1
    assert L.bbx_clipped2mask(None, 64, 64, None, None, None, 3, 3, 32, 64, 64, None, 12, 100.0, 2, None, None, None,
                              None, None, None, None) == -1
    assert L.bbx_set_option(None, 1, 1) == -1
    assert L.bbx_event_record(None, None) == -1 and L.bbx_event_query(None) == -1
    assert L.bbx_stream_wait_event(None, None) == -1
    assert L.bbx_copy_async(None, None, 16, 0, None) == -1
    assert L.bbx_funpack_tiles(None, 10, 10, 2, None, None, 1, None, None, None, 0, None, None) == -1


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'blackbox_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'bbx_oracle' not in src and 'import lacosmic' not in src and "'oracle'" not in src, f
    for f in ('blackbox.py',):
        p = os.path.join(ROOT, f)
        if os.path.isfile(p):
            assert 'oracle' not in open(p).read()
